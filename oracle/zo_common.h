/* oracle/zo_common.h -- shared helpers for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is a plain-C restatement of the zstd frame format (RFC 8878) and of libzstd 1.5.7's
 * level-3 ("dfast") encoder decisions as catalogued in SURVEY.md section 8(a) / Appendix A.
 * Nothing under python-zstandard_amd/ may include, link or call this code: it exists so tests,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have an in-repo checker that is itself
 * pinned against the real libzstd (oracle/_ref, built from /root/reference/zstd/zstd.c).
 */
#ifndef ZO_COMMON_H
#define ZO_COMMON_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* error codes: numeric values follow the reference's public enum (zstd/zstd_errors.h:61-97) so the
 * host layer can print the same strings (zstd.c:3580-3616). Functions return -(code). */
enum {
    ZO_OK = 0,
    ZO_E_GENERIC = 1,
    ZO_E_PREFIX_UNKNOWN = 10,
    ZO_E_FRAMEPARAM_UNSUPPORTED = 14,
    ZO_E_WINDOW_TOO_LARGE = 16,
    ZO_E_CORRUPTION = 20,
    ZO_E_CHECKSUM_WRONG = 22,
    ZO_E_LITERALS_HEADER_WRONG = 24,
    ZO_E_DICT_CORRUPTED = 30,
    ZO_E_DICT_WRONG = 32,
    ZO_E_PARAM_UNSUPPORTED = 40,
    ZO_E_TABLELOG_TOO_LARGE = 44,
    ZO_E_MAXSYMBOL_TOO_LARGE = 46,
    ZO_E_MAXSYMBOL_TOO_SMALL = 48,
    ZO_E_MEMORY = 64,
    ZO_E_DST_TOO_SMALL = 70,
    ZO_E_SRC_SIZE_WRONG = 72
};

#define ZO_MAGIC        0xFD2FB528u
#define ZO_DICT_MAGIC   0xEC30A437u
#define ZO_BLOCK_MAX    (1u << 17)
#define ZO_CONTENTSIZE_UNKNOWN ((uint64_t)-1)
#define ZO_CONTENTSIZE_ERROR   ((uint64_t)-2)

static inline uint16_t zo_rd16(const void* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t zo_rd24(const void* p) { const uint8_t* b = (const uint8_t*)p; return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16); }
static inline uint32_t zo_rd32(const void* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t zo_rd64(const void* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void zo_wr16(void* p, uint16_t v) { memcpy(p, &v, 2); }
static inline void zo_wr24(void* p, uint32_t v) { uint8_t* b = (uint8_t*)p; b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); }
static inline void zo_wr32(void* p, uint32_t v) { memcpy(p, &v, 4); }
static inline void zo_wr64(void* p, uint64_t v) { memcpy(p, &v, 8); }
static inline int zo_highbit(uint32_t v) { return 31 - __builtin_clz(v); } /* v != 0 */

/* ---- format constants (RFC 8878 section 3.1.1.3.2.1 and Appendix A default distributions) ---- */
#define ZO_MAXLL 35
#define ZO_MAXML 52
#define ZO_MAXOFF 31
#define ZO_LL_LOGMAX 9
#define ZO_ML_LOGMAX 9
#define ZO_OF_LOGMAX 8
#define ZO_LL_DEFLOG 6
#define ZO_ML_DEFLOG 6
#define ZO_OF_DEFLOG 5

extern const uint32_t zo_ll_base[36];
extern const uint8_t  zo_ll_bits[36];
extern const uint32_t zo_ml_base[53];
extern const uint8_t  zo_ml_bits[53];
extern const int16_t  zo_ll_defnorm[36];
extern const int16_t  zo_ml_defnorm[53];
extern const int16_t  zo_of_defnorm[29];

/* XXH64 (public algorithm; the frame checksum is its low 32 bits, zstd.c:28325-28329) */
uint64_t zo_xxh64(const void* data, size_t len, uint64_t seed);

/* FSE normalized-count header reader, shared by decoder and dictionary loader.
 * Returns bytes consumed or -(error). norm[] gets counts (-1 = "less than one"). */
int zo_fse_read_ncount(int16_t* norm, unsigned* maxSymbol, unsigned* tableLog,
                       const uint8_t* src, size_t srcSize);

int zo_huf_read_weights(uint8_t* w, unsigned* count, unsigned* log, const uint8_t* src, size_t srcSize);

#endif
