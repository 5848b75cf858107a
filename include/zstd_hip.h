/* include/zstd_hip.h -- C ABI of libzstd_hip.so, the MI355X (gfx950) backend for python-zstandard's
 * batch / one-shot frame hot path.
 *
 * Plain C, plain pointers and sizes, no torch / Python types. These entry points are what the reference's
 * C extension would bind in place of its file-local batch workers:
 *
 *   zhip_compress_batch    replaces  compress_from_datasources()   c-ext/compressor.c:1083-1336
 *                                    + compress_worker()           c-ext/compressor.c:856-1076
 *   zhip_decompress_batch  replaces  decompress_from_framesources() c-ext/decompressor.c:1185-1455
 *                                    + decompress_worker()          c-ext/decompressor.c:944-1181
 *   zhip_item              ==        DataSource / FramePointer      c-ext/compressor.c:805-808, decompressor.c:892-896
 *   zhip_segment           ==        BufferSegment                  c-ext/python-zstandard.h:307-313
 *   zhip_outbuf            ==        CompressorDestBuffer / DecompressorDestBuffer (compressor.c:816-821,
 *                                    decompressor.c:904-909): malloc()ed by the callee, ownership passes to the
 *                                    caller exactly like BufferWithSegments_FromMemory(useFree=1) expects
 *                                    (c-ext/bufferutil.c:107-148).
 *   error classes          ==        CompressorWorkerError / DecompressorWorkerError (compressor.c:823-828,
 *                                    decompressor.c:911-917) + first failing item index + zstd error code
 *                                    (numeric values of zstd/zstd_errors.h:61-97).
 *
 * The *_device entry points are the same operations with every buffer already resident in HBM
 * (what bench.py times and what a multi-GPU caller shards); the host-buffer entry points wrap them with
 * H2D / D2H copies. INTEGRATION.md shows the reference-side stub.
 */
#ifndef ZSTD_HIP_H
#define ZSTD_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZHIP_ABI_VERSION 3

typedef struct { uint64_t offset, length; } zhip_segment;             /* BufferSegment */
typedef struct { const void* src; size_t srcSize; size_t dstSize; } zhip_item; /* dstSize: decompress only (0 = unknown) */
typedef struct { void* data; size_t dataSize; zhip_segment* segs; size_t nSegs; } zhip_outbuf;

/* worker error classes (values follow the reference enums) */
enum {
    ZHIP_ERR_NONE = 0,
    ZHIP_ERR_ZSTD = 1,          /* zstdErr holds the zstd error code */
    ZHIP_ERR_NO_MEMORY = 2,
    ZHIP_ERR_SIZE_MISMATCH = 3, /* decompress: produced != expected (errDetail = produced, expected) ; compress: nospace */
    ZHIP_ERR_UNKNOWN_SIZE = 4,  /* decompress: frame has no content size and none supplied */
    ZHIP_ERR_HIP = 5,           /* HIP runtime failure; zhip_last_error() has the text */
    ZHIP_ERR_UNSUPPORTED = 6    /* parameter combination not implemented by this backend (fails loudly) */
};

typedef struct {
    int kind;            /* ZHIP_ERR_* */
    int zstdErr;         /* zstd error code when kind == ZHIP_ERR_ZSTD */
    size_t index;        /* first failing item */
    uint64_t detail[2];  /* size mismatch: produced, expected */
} zhip_error;

/* dictionary content type == ZSTD_dictContentType_e (zstd/zstd.h; ZstdCompressionDict(data, dict_type=...), c-ext/compressiondict.c:170-191,
 * consumed at c-ext/compressor.c:37-52 and c-ext/compressiondict.c:148-162): AUTO = a blob that starts with the dictionary magic is a
 * full dictionary, anything else raw content; RAWCONTENT = content whatever the first bytes are; FULLDICT = the magic is required
 * (compression reports "Dictionary mismatch", decompression "Dictionary is corrupted", as libzstd does). */
enum { ZHIP_DICT_AUTO = 0, ZHIP_DICT_RAWCONTENT = 1, ZHIP_DICT_FULLDICT = 2 };
/* frame format == ZSTD_format_e (ZstdCompressionParameters(format=...), ZstdDecompressor(format=...), c-ext/decompressor.c:32-35):
 * MAGICLESS frames have no 4-byte magic number in front of the frame header. */
enum { ZHIP_FORMAT_ZSTD1 = 0, ZHIP_FORMAT_ZSTD1_MAGICLESS = 1 };
/* explicit compression parameters == ZSTD_compressionParameters; the reference keeps them in the ZSTD_CCtx_params it applies to every
 * worker context (c-ext/compressor.c:20,203,1138; c-ext/compressionparams.c:13-120). 0 = "take it from the level" for each field,
 * exactly like ZSTD_CCtxParams_setParameter(…, 0). strategy: 1 = fast, 2 = dfast are the strategies this backend implements; a
 * combination that resolves to any other strategy fails loudly with ZHIP_ERR_UNSUPPORTED. */
typedef struct { uint32_t windowLog, chainLog, hashLog, searchLog, minMatch, targetLength; int32_t strategy; } zhip_compression_parameters;

/* compression parameters the hot path reads (the reference keeps them in ZSTD_CCtx_params,
 * c-ext/compressor.c:209-233; defaults there: contentSize=1, checksum=0, dictID=1). */
typedef struct {
    int level;
    int contentSizeFlag, checksumFlag, dictIDFlag;
    const void* dict; size_t dictSize;      /* raw bytes of a ZstdCompressionDict (or NULL) */
    int dictType;                           /* ZHIP_DICT_* */
    int format;                             /* ZHIP_FORMAT_* */
    zhip_compression_parameters cp;         /* all zero = derive everything from `level` */
} zhip_cparams;

typedef struct {
    const void* dict; size_t dictSize;      /* raw bytes of a ZstdCompressionDict (or NULL) */
    uint64_t maxWindowSize;                 /* 0 = default (1 << 27) */
    int dictType;                           /* ZHIP_DICT_* */
    int format;                             /* ZHIP_FORMAT_* */
} zhip_dparams;

/* ---- library / device ---- */
int         zhip_abi_version(void);
int         zhip_device_count(void);
int         zhip_set_device(int device);
const char* zhip_last_error(void);                  /* thread-local text of the last ZHIP_ERR_HIP */
const char* zhip_error_name(int zstdErr);            /* same strings as ZSTD_getErrorName (zstd.c:3580-3616) */
int         zhip_selftest(void);                     /* 0 = the wave primitives behave on this GPU */
size_t      zhip_compress_bound(size_t srcSize);     /* ZSTD_compressBound, zstd.h:249 */

/* ---- frame inspection (host, no GPU): ZSTD_getFrameContentSize zstd.c:43790, ZSTD_findFrameCompressedSize :44022 */
#define ZHIP_CONTENTSIZE_UNKNOWN ((uint64_t)-1)
#define ZHIP_CONTENTSIZE_ERROR   ((uint64_t)-2)
uint64_t zhip_frame_content_size(const void* src, size_t srcSize);
int64_t  zhip_find_frame_compressed_size(const void* src, size_t srcSize);  /* <0: -(zstd error code) */
/* the same with an explicit frame format (ZSTD_getFrameHeader_advanced zstd.c:43668, ZSTD_findFrameCompressedSize_advanced) */
uint64_t zhip_frame_content_size_format(const void* src, size_t srcSize, int format);
int64_t  zhip_find_frame_compressed_size_format(const void* src, size_t srcSize, int format);
/* level + size hints -> the parameters libzstd would use (ZSTD_getCParams zstd.c:30863; ZstdCompressionParameters.from_level,
 * c-ext/compressionparams.c:231-345). Host only. */
void     zhip_get_cparams(int level, uint64_t srcSizeHint, size_t dictSize, zhip_compression_parameters* out);
/* bytes of device memory the calling thread's contexts hold (scratch arenas, tables, staging; they grow with the largest batch seen and
 * are trimmed after very large ones) -- what ZstdCompressor.memory_size() / ZstdDecompressor.memory_size() report here, where the
 * reference reports ZSTD_sizeof_CCtx / ZSTD_sizeof_DCtx (c-ext/compressor.c:263, c-ext/decompressor.c:128). Creates the thread's context
 * (and its launch counters) if it does not exist yet, as the reference's contexts exist from the constructor on; 0 without a GPU. */
size_t   zhip_thread_memory_size(void);

/* ---- host-buffer batch API (drop-in for the reference's workers) ----
 * items are borrowed; *out is an array of *nOut buffers the caller owns -- one per pipeline chunk, like the reference's one per worker
 * (CompressorDestBuffer / DecompressorDestBuffer): release each `data` with zhip_free_payload() (large payloads are PINNED host
 * memory from a process-wide pool, so that the D2H copy lands in them at link speed; free() is correct only for the `segs` arrays),
 * each `segs` with free(), then the array with free() -- or everything with zhip_free_outbufs(bufs, n, 1).
 * Returns ZHIP_ERR_NONE or fills *err. Re-entrant; call with the GIL released. */
int  zhip_compress_batch(const zhip_cparams* params, const zhip_item* items, size_t n,
                         zhip_outbuf** out, size_t* nOut, zhip_error* err);
int  zhip_decompress_batch(const zhip_dparams* params, const zhip_item* items, size_t n, int requireSizes,
                           zhip_outbuf** out, size_t* nOut, zhip_error* err);
/* One call, every device (round 6). Both calls fan the batch out over the node's GPUs INSIDE the call, the way the reference fans it out over its worker
 * threads (c-ext/compressor.c:1127-1298, c-ext/decompressor.c:1237-1455): the item list is cut into contiguous runs of (almost) equal input bytes --
 * zhip_partition_by_bytes, the reference's rule -- every run goes to a host thread bound to its device (one persistent thread, context and staging area per
 * device slot), and the devices' zhip_outbufs come back in device order, which is item order; the first failing item (lowest index) is the one reported.
 * No collective is involved: the collection simply holds the buffers of every device. The environment selects the device slots: ZHIP_DEVICES=0,1,... (a
 * device may be listed twice: two contexts on one GPU); unset = every visible device, but at least ZHIP_DEVICE_MIN_BYTES (default 256 MiB) of input per
 * device, so small batches run on the calling thread's current device as they always did. Nothing changes for a caller. */
size_t zhip_partition_by_bytes(const uint64_t* sizes, size_t n, size_t workers, size_t* bounds /* 2 * workers: [start, end) per worker */);
int    zhip_batch_devices(int* devices, int cap);   /* the device slots the two calls above fan out over; returns their number */
void zhip_free_outbufs(zhip_outbuf* bufs, size_t n, int freePayload);
void zhip_free_payload(void* data);     /* a zhip_outbuf.data pointer: back to the pinned pool, or free() */

/* ---- device-resident batch API (all pointers are HBM addresses on the current device) ----
 * A context owns the per-launch scratch (literal buffers, hash tables, work counters, status words) so repeated
 * calls allocate nothing. `stream` is a hipStream_t passed as void* (NULL = default stream). Calls are asynchronous;
 * zhip_ctx_sync() waits and returns the first failing frame (lowest index) like the reference's workers do. */
typedef struct zhip_ctx zhip_ctx;
zhip_ctx* zhip_ctx_create(void);
void      zhip_ctx_destroy(zhip_ctx*);
int       zhip_ctx_set_ddict(zhip_ctx*, const void* hostDict, size_t dictSize, int dictType);   /* parses + uploads; NULL clears */
int       zhip_ctx_set_dformat(zhip_ctx*, int format, uint64_t maxWindowSize);     /* frame format + window limit of the next decode calls */
int       zhip_ctx_set_cparams(zhip_ctx*, const zhip_cparams* params);            /* uploads dict / tables */
/* What a device-API caller knows about the UNCOMPRESSED size of its largest item (0 = nothing; the host-buffer API sees the sizes itself).
 * Items above one block (128 KiB) are frames of several blocks. Told so, decompression runs the phase-split kernels in their several-block
 * mode (a block is the work item) and compression gives the generic kernel the whole chip -- or, in batches of thousands of such sources, the
 * flat match kernel searches them too. Untold, they are decoded / encoded one wave per frame by a token grid of the generic kernels: correct,
 * slow (the reference has no equivalent: its workers take any size, c-ext/compressor.c:1035). Frames of ONE block's size may also be frames of several
 * blocks -- libzstd's block splitter (levels 16 and up) cuts a 128 KiB source into many --: a decompress caller whose frames come from such levels
 * should say 131 073 or more here (the host-buffer API counts every frame's blocks itself and does). */
void      zhip_ctx_set_size_hint(zhip_ctx*, uint64_t maxItemBytes);

/* d_src: concatenated frames; d_srcSegs[i] = (offset,length) of frame i in d_src.
 * d_dst: output arena;        d_dstSegs[i] = (offset, capacity) where frame i must be written.
 * d_outSizes[i] receives the produced size, d_status[i] 0 or a zstd error code. */
int zhip_decompress_batch_device(zhip_ctx*, const void* d_src, const zhip_segment* d_srcSegs, size_t n,
                                 void* d_dst, const zhip_segment* d_dstSegs,
                                 uint64_t* d_outSizes, int32_t* d_status, void* stream);
/* d_dstSegs[i].length must be >= zhip_compress_bound(d_srcSegs[i].length). */
int zhip_compress_batch_device(zhip_ctx*, const void* d_src, const zhip_segment* d_srcSegs, size_t n,
                               void* d_dst, const zhip_segment* d_dstSegs,
                               uint64_t* d_outSizes, int32_t* d_status, void* stream);
int zhip_ctx_sync(zhip_ctx*, void* stream, const int32_t* d_status, size_t n, zhip_error* err);
/* The compress direction writes every frame into a zhip_compress_bound-sized slot; what is handed on (a BufferWithSegments, a payload
 * all-gatherv across GPUs) is the frames back to back. d_offsets[i] = where frame i goes inside d_dense (the caller's exclusive prefix
 * sum of d_outSizes, 8 bytes per item on the device); items whose d_status is non-zero are skipped. One wave per frame, 16 bytes per
 * lane. The collection step of c-ext/compressor.c:1407-1496 (per-worker destination buffers -> one result), on the device. */
int zhip_compact_device(const void* d_slots, const zhip_segment* d_slotSegs, const uint64_t* d_outSizes, const int32_t* d_status,
                        const uint64_t* d_offsets, size_t n, void* d_dense, void* stream);

/* name of a kernel as it appears in rocprofv3 traces ("" past the last one), and its average duration (ms) over the launches since the last call, measured with HIP events
 * on the stream it is launched on (for bench.py's roofline). k: 0 / 1 the generic decode / encode kernels, 2 K1 (with K0 and the bin pass in front of / behind it), 3 K2,
 * 4 K3, 5 / 6 the lane-serial match and the entropy kernels, 7 K1b -- which runs BESIDE K2 on a side stream: timed from K2's end to its own end, what it adds to the step --,
 * 8 the flat match kernel, 9 "zhip_decode_pipeline_span": not a kernel, a chunk's decode pipeline from K1's start to K3's end (what the overlapping kernels cost together). */
const char* zhip_kernel_name(int k);
int         zhip_ctx_kernel_time(zhip_ctx*, int direction, double* avgMs, uint64_t* launches);
/* the compress direction's table placement pick (zhip_compress_batch_device: the first launch of 16 384 frames or more -- 49 152 until round 6's last session -- times the match kernel on
 * up to three table allocations held side by side and keeps the fastest; where a probe is cheap -- dictionary batches -- and the three came out alike, up to three more):
 * ms3[k] = candidate k's PROBE time in ms for the first three (0 = not tried) -- since round 6's last session a probe launch searches the first 8 KiB of every source only, which ranks the
 * allocations like the whole launch does at a thirteenth of the time (30 against 36 ms per 65 536 sources of 128 KiB where the whole launches take 407 against 470). Returns the index kept (0..7: eight candidates are probed since round 6's last session -- one allocation in eight was a third, faster kind, 388-392 ms where the usual fast kind takes 406-417). */
int         zhip_ctx_table_pick(zhip_ctx*, float* ms3);

#ifdef __cplusplus
}
#endif
#endif
