/* oracle/zstd_oracle.h -- public entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libzstd_oracle.so.
 * Parity of this oracle is PINNED: tests/test_oracle_vs_golden.py checks it against the reference's own
 * golden vectors and against frames produced by the real libzstd 1.5.7
 * (oracle/_ref/libzstd_ref.so, built from /root/reference/zstd/zstd.c; fixtures under tests/golden/).
 */
#ifndef ZSTD_ORACLE_H
#define ZSTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t contentSize;   /* ZO_CONTENTSIZE_UNKNOWN if absent */
    uint64_t windowSize;
    uint32_t blockSizeMax;
    uint32_t headerSize;
    uint32_t dictID;
    uint32_t hasChecksum;
    uint32_t skippable;     /* 1: a skippable frame (magic 0x184D2A5?): headerSize = 8, windowSize = its payload size, contentSize = 0 */
} zo_frame_header;

/* decode side (zo_decode.c) -- negative return = -(zstd error code) */
int      zo_get_frame_header(zo_frame_header* h, const void* src, size_t srcSize);
uint64_t zo_frame_content_size(const void* src, size_t srcSize);
int64_t  zo_find_frame_compressed_size(const void* src, size_t srcSize);
int64_t  zo_decompress_frame(void* dst, size_t dstCap, const void* src, size_t srcSize,
                             const void* dict, size_t dictSize, size_t* srcConsumed);

/* encode side (zo_encode.c): one frame, libzstd-1.5.7-bit-exact for the supported strategies.
 * flags: bit0 contentSize, bit1 checksum, bit2 dictID (python-zstandard defaults: 1,0,1). */
#define ZO_F_CONTENTSIZE 1
#define ZO_F_CHECKSUM    2
#define ZO_F_DICTID      4
size_t   zo_compress_bound(size_t srcSize);
int64_t  zo_compress_frame(void* dst, size_t dstCap, const void* src, size_t srcSize,
                           int level, unsigned flags, const void* dict, size_t dictSize);

uint64_t zo_xxh64(const void* data, size_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
