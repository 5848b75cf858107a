// zhip_lib.hip -- libzstd_hip.so: C ABI (include/zstd_hip.h) + kernel entry points, compiled for gfx950 only.
//
// Host side = the batch dispatcher that replaces compress_from_datasources / decompress_from_framesources
// (c-ext/compressor.c:1083, c-ext/decompressor.c:1185): instead of partitioning frames over a pthread pool
// (POOL_*, zstd.c:7537-7975) it stages them in HBM and launches persistent one-wave workgroups that pull frame
// indices from an atomic counter.
#include <atomic>
#include <chrono>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/zstd_hip.h"
#include "zhip_decode_pipeline.hpp"
#include "zhip_encode_kernel.hpp"
#include "zhip_cparams.hpp"

#define ZHIP_LDS_BYTES (160u * 1024u)      // per CU on gfx950
#ifndef ZHIP_E1LDS_PER_CU
#define ZHIP_E1LDS_PER_CU 4                // batches up to this many frames per CU take the LDS-source match kernel (ZHIP_E1LDS_MAX overrides the frame count); r02zq: 1 024 frames 147 ms against 159-163 with the flat kernel, 512: 89 against 146
#endif
static_assert(sizeof(ZpSeqQLDS) <= ZHIP_LDS_BYTES && sizeof(ZpHufKernelLDS) <= ZHIP_LDS_BYTES, "a workgroup's LDS must fit a CU");

// ------------------------------------------------------------------------------------------ kernels
ZH_GLOBAL __launch_bounds__(64, 2) void zhip_decode_frames_kernel(ZhipDecodeArgs a)
{
    __shared__ ZdLDS L;
    zd_kernel_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64, 2) void zhip_decode_lit_kernel(ZhipPipeArgs a)
{
    __shared__ ZdLDS L;
    zp_lit_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_lit_lanes_kernel(ZhipPipeArgs a) { zp_lit_lanes_body(a); }      // K1's lane-per-frame pass over dictionary batches
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_check_kernel(ZhipPipeArgs a) { zp_check_body(a); }      // KX: content checksums, a lane per frame, after K3
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_pre_kernel(ZhipPipeArgs a)           // K0: a lane per frame walks what K1's lane 0 used to (Huffman weights, sequence distributions)
{
    __shared__ ZpPreLDS L;
    zp_pre_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_bin_kernel(ZhipPipeArgs a)
{
    __shared__ ZpBinLDS L;
    zp_bin_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_huf_kernel(ZhipPipeArgs a)
{
    __shared__ ZpHufKernelLDS L;
    zp_huf_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_seq_kernel(ZhipPipeArgs a)           // K2: a quad of lanes per frame, four waves per CU
{
    __shared__ ZpSeqQLDS L;
    zp_seqq_body<false>(a, L);
}
// ---- frames of several blocks (ZpFrameRec in zhip_format.hpp): K1 per frame over its blocks, K2 per block, K3 per frame over its blocks
ZH_GLOBAL __launch_bounds__(64, 2) void zhip_decode_lit_mb_kernel(ZhipPipeArgs a)
{
    __shared__ ZdLDS L;
    zp_lit_mb_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_decode_seq_mb_kernel(ZhipPipeArgs a)
{
    __shared__ ZpSeqQLDS L;
    zp_seqq_body<true>(a, L);
}
#ifndef ZP_K3_MINWAVES
#define ZP_K3_MINWAVES 6         // 77 VGPRs: own-lane items <= 16 bytes (zhip_decode_pipeline.hpp: ZP_LIT_SHORT / ZP_FAR_SHORT, r03g: 113 -> 96) and the phase timers
                                 // in their own instantiation (r03n: 96 -> 77). Six waves per SIMD; seven (72 VGPRs) and eight measured the same
#endif
ZH_GLOBAL __launch_bounds__(64, ZP_K3_MINWAVES) void zhip_decode_exec_kernel(ZhipPipeArgs a)
{
    __shared__ ZpExecLDS L;
    zp_exec_body<false, false>(a, L);
}
ZH_GLOBAL __launch_bounds__(64, 4) void zhip_decode_exec_prof_kernel(ZhipPipeArgs a)      // ZHIP_PROF=1: the same kernel with its phase timers
{
    __shared__ ZpExecLDS L;
    zp_exec_body<false, true>(a, L);
}
#ifndef ZP_K3D_MINWAVES
#define ZP_K3D_MINWAVES 4        // 113 VGPRs since round 3 (round 2: 168 at three waves per SIMD, 208 bytes of spills at four, r02y)
#endif
ZH_GLOBAL __launch_bounds__(64, ZP_K3D_MINWAVES) void zhip_decode_exec_dict_kernel(ZhipPipeArgs a)      // the context has a dictionary
{
    __shared__ ZpExecLDS L;
    zp_exec_body<true, false>(a, L);
}
ZH_GLOBAL __launch_bounds__(64, 4) void zhip_decode_exec_mb_kernel(ZhipPipeArgs a)
{
    __shared__ ZpExecLDS L;
    zp_exec_body<false, false, true>(a, L);
}
ZH_GLOBAL __launch_bounds__(64, 4) void zhip_decode_exec_mb_dict_kernel(ZhipPipeArgs a)
{
    __shared__ ZpExecLDS L;
    zp_exec_body<true, false, true>(a, L);
}
ZH_GLOBAL __launch_bounds__(64, 3) void zhip_encode_frames_kernel(ZhipEncodeArgs a)
{
    __shared__ ZeLDS L;
    __shared__ ZeLDSMulti M;
    ze_kernel_body(a, L, M);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_match_kernel(ZhipEncodeArgs a) { ze_match_body(a); }
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_split_kernel(ZhipEncodeArgs a)          // block layout of sources of several blocks for the flat match kernel
{
    __shared__ ZeLDS L;
    ze_split_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_trailer_kernel(ZhipEncodeArgs a) { ze_trailer_body(a); }      // EX: checksum trailers, a lane per frame, after E2
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_match_flat_kernel(ZhipEncodeArgs a) { ze_match_flat_body<2>(a); }
// four probes per trip: chunks small enough to be bound by a source's serial chain rather than by the memory system (ze_dfast_flat_np)
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_match_flat4_kernel(ZhipEncodeArgs a) { ze_match_flat_body<4>(a); }
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_match_flat3_kernel(ZhipEncodeArgs a) { ze_match_flat_body<3>(a); }      // three probes: launches of up to 65 536 sources
ZH_GLOBAL __launch_bounds__(64) void zhip_encode_match_flat_mb_kernel(ZhipEncodeArgs a) { ze_match_flat_mb_body(a); }
static_assert(sizeof(ZeSrcLDS<ZF_BLOCK_MAX>) <= ZHIP_LDS_BYTES, "a workgroup's LDS must fit a CU");
template <uint32_t BYTES, int NPROBE> __global__ __launch_bounds__(64) void zhip_encode_match_lds_kernel(ZhipEncodeArgs a)
{
    __shared__ ZeSrcLDS<BYTES> L;
    ze_match_lds_body<NPROBE>(a, L.b, BYTES);
}
#ifndef ZE_E2_MINWAVES
#define ZE_E2_MINWAVES 4
#endif
ZH_GLOBAL __launch_bounds__(64, ZE_E2_MINWAVES) void zhip_encode_entropy_kernel(ZhipEncodeArgs a)
{
    __shared__ ZeLDS L;
    ze_entropy_body(a, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_build_cdict_kernel(const uint8_t* dict, uint32_t dictSize, const ZhipDictEntropy* de, ZeRows rows,
                                                              ZeCDict* cd, uint32_t* hashLong, uint32_t* hashSmall, uint32_t* tmpLong)
{
    __shared__ ZeLDS L;
    ze_cdict_body(dict, dictSize, de, rows, cd, hashLong, hashSmall, tmpLong, L);
}
ZH_GLOBAL void zhip_selftest_kernel(uint32_t* out)
{
    uint32_t v = zh_scan_add(zh_lane());                  // 0+1+..+lane
    uint64_t m = zh_ballot((zh_lane() & 1) != 0);
    out[zh_lane()] = v + (uint32_t)zh_popc64(m) + zh_shfl(zh_lane(), 63) + zh_first(zh_lane() + 7);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_dict_tables_kernel(const ZhipDictEntropy* de, ZhipDictTables* out)
{
    __shared__ ZdLDS L;
    zp_dict_tables_body(de, out, L);
}
ZH_GLOBAL __launch_bounds__(64) void zhip_parse_dict_kernel(const uint8_t* dict, uint32_t dictSize, ZhipDictEntropy* de)
{
    __shared__ ZdLDS L;
    zd_dict_body(dict, dictSize, de, L);
}

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_lastError;
static int hip_fail(hipError_t e, const char* what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_lastError = buf;
    return ZHIP_ERR_HIP;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

extern "C" const char* zhip_last_error(void) { return g_lastError.c_str(); }
extern "C" int zhip_abi_version(void) { return ZHIP_ABI_VERSION; }
extern "C" int zhip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { hip_fail(e, "hipGetDeviceCount"); return 0; }
    return n;
}
extern "C" int zhip_set_device(int d) { HIP_TRY(hipSetDevice(d)); return 0; }
extern "C" size_t zhip_compress_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }

extern "C" const char* zhip_error_name(int code)
{
    switch (code) {   // strings of ERR_getErrorString, zstd.c:3580-3616
    case ZE_OK: return "No error detected";
    case ZE_GENERIC: return "Error (generic)";
    case ZE_PREFIX_UNKNOWN: return "Unknown frame descriptor";
    case ZE_FRAMEPARAM_UNSUPPORTED: return "Unsupported frame parameter";
    case ZE_WINDOW_TOO_LARGE: return "Frame requires too much memory for decoding";
    case ZE_CORRUPTION: return "Data corruption detected";
    case ZE_CHECKSUM_WRONG: return "Restored data doesn't match checksum";
    case ZE_LITERALS_HEADER_WRONG: return "Header of Literals' block doesn't respect format specification";
    case ZE_DICT_CORRUPTED: return "Dictionary is corrupted";
    case ZE_DICT_WRONG: return "Dictionary mismatch";
    case ZE_PARAM_UNSUPPORTED: return "Unsupported parameter";
    case ZE_PARAM_OUTOFBOUND: return "Parameter is out of bound";
    case ZE_TABLELOG_TOO_LARGE: return "tableLog requires too much memory : unsupported";
    case ZE_MAXSYMBOL_TOO_LARGE: return "Unsupported max Symbol Value : too large";
    case ZE_MAXSYMBOL_TOO_SMALL: return "Specified maxSymbolValue is too small";
    case ZE_MEMORY: return "Allocation error : not enough memory";
    case ZE_DST_TOO_SMALL: return "Destination buffer is too small";
    case ZE_SRC_SIZE_WRONG: return "Src size is incorrect";
    default: return "Unspecified error code";
    }
}

// ------------------------------------------------------------------------------------------ frame inspection (host)
static inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t rd24(const uint8_t* p) { return rd16(p) | ((uint32_t)p[2] << 16); }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct HostFrameHeader { uint64_t contentSize; uint32_t headerSize; uint32_t hasChecksum; uint32_t skippable, skipSize; };
// RFC 8878 3.1.1.1; restates ZSTD_getFrameHeader_advanced (zstd.c:43668) for the fields the dispatcher needs.
static int host_frame_header(HostFrameHeader* h, const uint8_t* src, size_t n, int format)
{
    const size_t mg = format == ZHIP_FORMAT_ZSTD1_MAGICLESS ? 0 : 4;
    h->skippable = 0; h->skipSize = 0;
    if (n < mg + 1) return -ZE_SRC_SIZE_WRONG;
    if (mg && (rd32(src) & 0xFFFFFFF0u) == ZF_MAGIC_SKIPPABLE) {          // skippable frame: content size 0 (ZSTD_getFrameContentSize, zstd.c:43773)
        if (n < 8) return -ZE_SRC_SIZE_WRONG;
        h->skippable = 1; h->skipSize = rd32(src + 4); h->contentSize = 0; h->headerSize = 8; h->hasChecksum = 0;
        return 0;
    }
    if (mg && rd32(src) != ZF_MAGIC) return -ZE_PREFIX_UNKNOWN;
    src += mg; n -= mg;
    uint32_t fhd = src[0], dictCode = fhd & 3, single = (fhd >> 5) & 1, fcsCode = fhd >> 6;
    uint32_t dictBytes = dictCode == 3 ? 4 : dictCode, fcsBytes = fcsCode == 0 ? single : (1u << fcsCode);
    uint32_t hs = 1 + (single ? 0 : 1) + dictBytes + fcsBytes;
    if (fhd & 8) return -ZE_FRAMEPARAM_UNSUPPORTED;
    if (n < hs) return -ZE_SRC_SIZE_WRONG;
    const uint8_t* p = src + 1 + (single ? 0 : 1) + dictBytes;
    h->contentSize = ZHIP_CONTENTSIZE_UNKNOWN;
    if (fcsCode == 0) { if (single) h->contentSize = p[0]; }
    else if (fcsCode == 1) h->contentSize = (uint64_t)rd16(p) + 256;
    else if (fcsCode == 2) h->contentSize = rd32(p);
    else h->contentSize = rd64(p);
    h->headerSize = (uint32_t)mg + hs; h->hasChecksum = (fhd >> 2) & 1;
    return 0;
}
// how many blocks a frame has (0: not a frame this walk can follow -- the kernels will say what is wrong with it). The host-buffer decompress call uses it to tell frames of
// SEVERAL blocks from frames of one: libzstd's block splitter (levels 16 and up) cuts even a 128 KiB source into many, and those belong to the pipeline's several-block mode,
// not -- one wave each -- to the generic kernel (round 6: 2 048 level-19 frames of 128 KiB took 44 ms there, 6 GB/s)
static uint32_t host_count_blocks(const uint8_t* src, size_t n, int format)
{
    HostFrameHeader h;
    if (host_frame_header(&h, src, n, format) < 0 || h.skippable) return 0;
    size_t pos = h.headerSize; uint32_t nb = 0;
    for (;;) {
        if (pos + 3 > n) return 0;
        const uint32_t bh = rd24(src + pos); pos += 3;
        const uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
        if (type == 3) return 0;
        const size_t body = type == 1 ? 1 : bs;
        if (pos + body > n) return 0;
        pos += body; nb++;
        if (bh & 1) return nb;
        if (nb > 65535) return 0;
    }
}
extern "C" uint64_t zhip_frame_content_size_format(const void* src, size_t n, int format)
{
    HostFrameHeader h;
    if (host_frame_header(&h, (const uint8_t*)src, n, format) < 0) return ZHIP_CONTENTSIZE_ERROR;
    return h.contentSize;
}
extern "C" uint64_t zhip_frame_content_size(const void* src, size_t n) { return zhip_frame_content_size_format(src, n, ZHIP_FORMAT_ZSTD1); }
extern "C" int64_t zhip_find_frame_compressed_size_format(const void* srcv, size_t n, int format)
{
    const uint8_t* src = (const uint8_t*)srcv;
    HostFrameHeader h; int e = host_frame_header(&h, src, n, format); if (e < 0) return e;
    if (h.skippable) return (uint64_t)h.skipSize + 8 > n ? -ZE_SRC_SIZE_WRONG : (int64_t)h.skipSize + 8;      // readSkippableFrameSize (zstd.c:43795)
    size_t pos = h.headerSize;
    for (;;) {
        if (pos + 3 > n) return -ZE_SRC_SIZE_WRONG;
        uint32_t bh = rd24(src + pos); pos += 3;
        uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
        if (type == 3) return -ZE_CORRUPTION;
        size_t c = type == 1 ? 1 : bs;
        if (pos + c > n) return -ZE_SRC_SIZE_WRONG;
        pos += c;
        if (bh & 1) break;
    }
    if (h.hasChecksum) { if (pos + 4 > n) return -ZE_SRC_SIZE_WRONG; pos += 4; }
    return (int64_t)pos;
}
extern "C" int64_t zhip_find_frame_compressed_size(const void* src, size_t n) { return zhip_find_frame_compressed_size_format(src, n, ZHIP_FORMAT_ZSTD1); }
extern "C" void zhip_get_cparams(int level, uint64_t srcSizeHint, size_t dictSize, zhip_compression_parameters* out) { if (out) zh_get_cparams(level, srcSizeHint, dictSize, out); }

// ------------------------------------------------------------------------------------------ context
#ifndef ZHIP_NSLOT
#define ZHIP_NSLOT 3
#endif
#ifndef ZHIP_E1LDS_PROBES
#define ZHIP_E1LDS_PROBES 2       // probes per trip of the LDS-source match kernel (small batches: one source per CU)
#endif
#ifndef ZHIP_TABLE_EPOCHS
#define ZHIP_TABLE_EPOCHS 1      // launch numbers in the cells of the flat searches' tables; 0: the tables are zeroed every launch -- a memset, the dictionary search's own waves -- (A/B build)
#endif
#ifndef ZHIP_FAST_WIDE
#define ZHIP_FAST_WIDE 1          // fast-strategy batches above 32 768 sources: 65 536 per chunk at sixteen sources per wave; 0: chunks of 32 768 at eight (A/B build)
#endif
#ifndef ZHIP_TRAILER_LATER
#define ZHIP_TRAILER_LATER 1     // compress: checksum trailers by EX after the entropy kernel; 0: by the entropy kernel on one lane per frame (A/B build)
#endif
#ifndef ZHIP_SIDE
#define ZHIP_SIDE 1              // K1b beside K2 on a side stream; 0: the decode kernels one after the other on one stream, two chunk slots (rounds 1-5; A/B build)
#endif
#ifndef ZHIP_PICK_MIN
#define ZHIP_PICK_MIN 16384       // sources per launch from which a context's first flat-search launch picks its tables' placement (49 152 while a probe was a whole launch: with probes at ~9 ms per
                                  // 16 384 sources the host-buffer API's chunks of 32 768 pick too -- multi_compress_to_buffer of 65 536 x 128 KiB 12.4-13.5 -> 13.5-13.9 GB/s in five alternating pairs, r06zzt)
#endif
#ifndef ZHIP_PICK_CANDIDATES
#define ZHIP_PICK_CANDIDATES 8     // table allocations the placement pick probes (3, up to 6 when alike, until r06zzv)
#endif
#ifndef ZHIP_PICK_STUDY
#define ZHIP_PICK_STUDY 0        // 1: DIAGNOSTIC build -- the placement pick also times eight candidate allocations whole and over the sources' first bytes, and prints them
#endif
#ifndef ZHIP_K0
#define ZHIP_K0 1                // K0 (zhip_decode_pre_kernel) in front of K1; 0: K1 parses every description itself (A/B build)
#endif
#ifndef ZHIP_DCHUNK
#define ZHIP_DCHUNK 65536        // frames per chunk of the decode pipeline (r02zl: 301 GB/s in one 65 536-frame chunk against 297 in two of 32 768: longer launches amortise their tails, and the two chunk slots overlap little anyway)
#endif
static thread_local int g_reserveRc = ZHIP_ERR_HIP;      // why the last failed DevBuf::reserve failed (ZHIP_ERR_NO_MEMORY or ZHIP_ERR_HIP)
// (device bytes are accounted per CONTEXT -- zhip_ctx::device_bytes sums its buffers -- not in a thread-local counter: a context may be
// destroyed, or grown, on another thread than the one that created it, and a counter that is decremented where it was never incremented
// wraps)
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    uint64_t gen = 0;            // which allocation this is, process-wide (a released buffer's address may come back with other contents: whoever remembers something ABOUT the contents remembers this too)
    int reserve(size_t n) {
        if (n <= cap) return 0;
        release();
        size_t want = n + (n >> 3) + 4096;
        if (want < n) { g_lastError = "allocation size overflow"; return g_reserveRc = ZHIP_ERR_NO_MEMORY; }
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); p = nullptr; g_lastError = "out of device memory"; return g_reserveRc = ZHIP_ERR_NO_MEMORY; }
        if (e != hipSuccess) { p = nullptr; return g_reserveRc = hip_fail(e, "hipMalloc"); }
        static std::atomic<uint64_t> g_gen{0};
        cap = want; gen = ++g_gen; return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};
#define ZHIP_NTIMER 10
struct KTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // owned: destroyed after reading
    std::vector<std::pair<hipEvent_t, hipEvent_t>> shared;    // borrowed: another timer owns the events
    double totalMs = 0; uint64_t launches = 0;
};

struct zhip_ctx {
    int device = 0;
    int numCU = 0;
    int decBlocksPerCU = 0;
    int encBlocksPerCU = 0;
    int k1PerCU = 0, k2PerCU = 0, k3PerCU = 0;
    DevBuf pipeMeta, pipeLit, pipeCounters, pipeFallback, pipeFse, pipeOrder, pipeHuf, pipeOrderLit, pipeItemFrame, pipeItemReps, pipeFrameRecs, pipeBases, pipePre;
    hipStream_t slotStream[ZHIP_NSLOT] = {};
    hipStream_t sideStream[ZHIP_NSLOT] = {};
    DevBuf encWorkspace, encMeta, encArena, encTables, encBigList, encBigWs, encFlatTables, encE1List, encMbBlocks, encMbCount, encMbSeqs;
    int e1PerCU = 0, e2PerCU = 0;
    size_t srcMaxHint = 0;             // largest source of the batch being launched when the caller knows it (host-buffer API), else 0
    size_t dstMaxHint = 0;             // the same for the decode direction: largest announced content size of the batch (host-buffer API), else 0
    bool hostPipe = false;             // the host-buffer pipeline is the caller (decompress_batch_one): its chunks' kernels are far from its bound -- the copies over the link -- and it
                                       // already keeps three streams busy; the decode step's side stream only added a queue for those copies to share (65 536 x 128 KiB inside a process
                                       // with other streams: 46-48 -> 38-43 GB/s with it, r06zl), so K1b stays on the chunk's stream there
    size_t dstSlotsHint = 0;           // decode, host-buffer API: block slots the frames of the batch being launched are expected to need (1 per frame of one block, 2 per 128 KiB + 2 above), else 0
    size_t itemHint = 0;               // zhip_ctx_set_size_hint: what a device-API caller says about its items' uncompressed sizes (0 = nothing)
    zhip_cparams cparams = {3, 1, 0, 1, nullptr, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
    ZeRows rows = {};                  // cparams resolved per source-size class (zhip_cparams.hpp)
    DevBuf scratch, counter;
    // dictionary (compress side): raw bytes, parsed entropy section, digested form and its tagged tables
    DevBuf cdictBlob, cdictEntropy, cdictDigest, cdictTables;
    bool hasCDict = false; uint32_t cdictContentOffset = 0, cdictAttachMax = ZE_DICT_ATTACH_MAX; int cdictHlog = 0, cdictClog = 0, cdictStrat = 2;
    uint32_t cdictContentSize = 0;
    // launch numbers in the cells of the flat dictionary search's tables (ZhipEncodeArgs.tabEpoch): the last one used on the allocation at encEpochPtr, the index width and
    // dictionary it was counted for -- any of them changing, or the numbers running out, zeroes the allocation and starts again at 1
    uint32_t encEpoch = 0, encEpochShift = 0; void* encEpochPtr = nullptr; uint64_t encEpochKey = 0, encEpochGen = 0; size_t encEpochBytes = 0;      // (encEpochBytes: how much of the allocation has been zeroed -- a larger batch that still fits the allocation reaches slots that never were)
    uint64_t cdictKey = 0, ddictKey = 0;     // fingerprint of the dictionary currently digested (skip re-digesting per call)
    // dictionary (decode side)
    DevBuf dictBlob, dictEntropy, dictTables;
    uint32_t dictSize = 0, dictID = 0, dictContentOffset = 0; bool dictHasEntropy = false;
    uint64_t maxWindowSize = (1ull << 27) + 1;
    int dformat = ZHIP_FORMAT_ZSTD1;
    // host-API staging
    DevBuf hSrc, hDst, hSegs, hStatus, hDense;
    void* pinned = nullptr; size_t pinnedCap = 0;
    bool hpReady = false; hipStream_t hpH2D = nullptr, hpCompute = nullptr, hpD2H = nullptr;     // host pipeline: copy-in, kernels, copy-out
    void* hpStage[2] = {nullptr, nullptr}; size_t hpStageCap[2] = {0, 0}; hipEvent_t hpStageFree[2] = {nullptr, nullptr}; int hpNextSlot = 0;
    // bring-up / tuning knobs, read from the environment ONCE when the context is created (never in a launch path)
    struct Knobs {
        // bring-up aids
        bool prof = false;                  // ZHIP_PROF: the kernels' phase timers (a separate instantiation of K3; printed to stderr)
        size_t k0Min = 6144;                // ZHIP_K0_MIN: frames per chunk from which K0 runs in front of K1 (a GPU test sets 0: K0 on every batch)
        // decode pipeline
        size_t dchunk = ZHIP_DCHUNK;        // frames (several-block mode: block slots) per chunk
        int nslot = 2; bool nslotSet = false;   // chunk slots on their own streams (small frames get a third unless ZHIP_NSLOT says otherwise)
        int k1PerCU = 0, k3PerCU = 0;       // waves per CU of K1 / K3 (0: what the occupancy query says)
        // encode
        size_t echunk = 0;                  // ZHIP_ECHUNK: sources per chunk, upper bound (0: none)
        size_t echunkMax = 0;               // ZHIP_ECHUNK_MAX: frames per launch of the flat match kernel (0 = 65 536, and 131 072 for larger batches where memory allows)
        long e1LdsMax = -1; size_t e1LdsRounds = 2;     // the LDS-source match kernel of small batches: most sources it takes (-1: by the CU count), rounds per CU
        size_t mbcMin = 4096;               // sources per 256 KiB of size hint from which sources of several blocks take the flat search (8 192 until r06zzi)
        unsigned mbcLanes = 32;             // sources per wave of that search (64 / 32 / 16 / 8 within 10-30 % of each other, r03z)
        // probes per trip of the flat double-fast search by launch size: up to flat4Max sources four (bound by one source's serial chain: 21-25 % less time from
        // 1 024 to 32 768 sources, r04zd), up to flat3Max three, above two (with the placement picked, at 65 536: 421 / 415 / 425 ms for two / three / four, r05w)
        size_t flat4Max = 32768, flat3Max = 65536; bool flat3 = true;
        bool e1fPick = true;                // ZHIP_E1F_PICK=0: take the flat tables where the first allocation put them (zhip_compress_batch_device)
        // host-buffer pipeline
        size_t hchunkE = 32768, hchunkE0 = 0;   // compress: items per chunk, items of the first chunk (0: like the others)
        size_t hchunkD0 = 2048;             // decompress: items of the first chunk of batches above 8 192 (0: like the others)
        unsigned packThreads = 0;           // host threads packing items into the pinned staging slots (0: ZHIP_PACK_THREADS)
    } knob;
    size_t flatMaxCached = 0;                    // frames per launch of the flat match kernel for batches above 65 536 (0: not decided yet)
    bool e1fPicked = false; float e1fPickMs[3] = {0, 0, 0}; int e1fPickKept = 0; void* e1fPickedPtr = nullptr;      // the flat match kernel's tables: placement picked once per context (zhip_compress_batch_device)
    bool timing = false;                         // per-kernel HIP-event timers: off until zhip_ctx_kernel_time() is first called
    unsigned long long* profDecode = nullptr;    // ZHIP_PROF phase-timer accumulators, owned by the context (one context == one caller)
    unsigned long long* profPipe = nullptr;
    unsigned long long* profEncode = nullptr;
    KTimer timer[ZHIP_NTIMER];   // 0 fused decode, 1 fused encode, 2 K1 literals, 3 K2 sequences, 4 K3 execution, 5 E1 match, 6 E2 entropy, 7 K1b Huffman streams, 8 the flat match kernel, 9 the decode pipeline of a chunk from K1's start to K3's end (K1b runs beside K2)
    size_t device_bytes() const
    {
        const DevBuf* all[] = {&pipeMeta, &pipeLit, &pipeCounters, &pipeFallback, &pipeFse, &pipeOrder, &pipeHuf, &pipeOrderLit, &pipeItemFrame, &pipeItemReps, &pipeFrameRecs, &pipeBases, &pipePre, &encWorkspace, &encMeta, &encArena,
                               &encTables, &encBigList, &encBigWs, &encFlatTables, &encE1List, &encMbBlocks, &encMbCount, &encMbSeqs, &scratch, &counter, &cdictBlob, &cdictEntropy, &cdictDigest, &cdictTables,
                               &dictBlob, &dictEntropy, &dictTables, &hSrc, &hDst, &hSegs, &hStatus, &hDense};
        size_t n = 0;
        for (const DevBuf* b : all) n += b->cap;
        return n;
    }
};

extern "C" zhip_ctx* zhip_ctx_create(void)
{
    zhip_ctx* c = new zhip_ctx();
    if (hipGetDevice(&c->device) != hipSuccess) { g_lastError = "hipGetDevice failed (no GPU?)"; delete c; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { g_lastError = "hipGetDeviceProperties failed"; delete c; return nullptr; }
    c->numCU = prop.multiProcessorCount;
    {   zhip_ctx::Knobs& k = c->knob;
        // Run-time knobs (round 6: nine, each exercised by a GPU test or a resource policy; rounds 1-5's other twenty-two -- chunk shapes, probe counts, waves per CU,
        // bring-up aids -- are the constants of this struct, and what lost its A/B is in DESIGN.md with its measurements): ZHIP_PROF (phase timers), ZHIP_K0_MIN, ZHIP_E1F_PICK,
        // ZHIP_E1LDS_MAX, ZHIP_MBC_MIN here; ZHIP_DEVICES / ZHIP_DEVICE_MIN_BYTES (the in-call device fan-out), ZHIP_KEEP_GB, ZHIP_PIN_POOL_KEEP_MB (memory kept between
        // calls) where they are used.
        k.prof = getenv("ZHIP_PROF") != nullptr;
        if (const char* e = getenv("ZHIP_K0_MIN")) k.k0Min = (size_t)atol(e);
        if (const char* e = getenv("ZHIP_MBC_MIN")) k.mbcMin = (size_t)atol(e);     // compress: batches of at least this many sources take the flat search for sources of several blocks
        if (const char* e = getenv("ZHIP_E1LDS_MAX")) { const long v = atol(e); if (v >= 0 && v <= 65536) k.e1LdsMax = v; }
        if (const char* e = getenv("ZHIP_E1F_PICK")) k.e1fPick = atol(e) != 0;    // 0: take the tables where the first allocation put them (A/B)
    }
    zh_resolve_rows(&c->rows, 3, nullptr);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_decode_frames_kernel, 64, 0) != hipSuccess || nb < 1) nb = 8;
    c->decBlocksPerCU = nb;
    nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_encode_frames_kernel, 64, 0) != hipSuccess || nb < 1) nb = 4;
    c->encBlocksPerCU = nb;
    nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_decode_lit_kernel, 64, 0) != hipSuccess || nb < 1) nb = 8;
    c->k1PerCU = nb;
    nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_decode_seq_kernel, 64, 0) != hipSuccess || nb < 1) nb = 3;
    c->k2PerCU = nb;
    nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_decode_exec_kernel, 64, 0) != hipSuccess || nb < 1) nb = 16;
    c->k3PerCU = nb;
    nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_encode_match_kernel, 64, 0) != hipSuccess || nb < 1) nb = 4;
    c->e1PerCU = nb;
    nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zhip_encode_entropy_kernel, 64, 0) != hipSuccess || nb < 1) nb = 4;
    c->e2PerCU = nb;
    return c;
}
static void drain_shared(KTimer& t)
{
    for (auto& pr : t.shared) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { t.totalMs += ms; t.launches++; }
    }
    t.shared.clear();
}
static void drain_timer(KTimer& t)
{
    drain_shared(t);
    for (auto& pr : t.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { t.totalMs += ms > 0 ? ms : 0; t.launches++; }
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    t.pending.clear();
}
extern "C" void zhip_ctx_destroy(zhip_ctx* c)
{
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < ZHIP_NTIMER; i++) drain_shared(c->timer[i]);
    for (int i = 0; i < ZHIP_NTIMER; i++) drain_timer(c->timer[i]);
    for (int i = 0; i < ZHIP_NSLOT; i++) if (c->slotStream[i]) (void)hipStreamDestroy(c->slotStream[i]);
    for (int i = 0; i < ZHIP_NSLOT; i++) if (c->sideStream[i]) (void)hipStreamDestroy(c->sideStream[i]);
    c->pipeMeta.release(); c->pipeLit.release(); c->pipeCounters.release(); c->pipeFallback.release(); c->pipeFse.release(); c->pipeOrder.release(); c->pipeHuf.release(); c->pipeOrderLit.release(); c->pipeItemFrame.release(); c->pipeItemReps.release(); c->pipeFrameRecs.release(); c->pipeBases.release(); c->pipePre.release();
    c->scratch.release(); c->counter.release(); c->encWorkspace.release(); c->encMeta.release(); c->encArena.release(); c->encTables.release(); c->encBigList.release(); c->encBigWs.release(); c->encFlatTables.release(); c->encE1List.release(); c->encMbBlocks.release(); c->encMbCount.release(); c->encMbSeqs.release(); c->dictBlob.release(); c->dictEntropy.release(); c->dictTables.release();
    c->cdictBlob.release(); c->cdictEntropy.release(); c->cdictDigest.release(); c->cdictTables.release();
    c->hSrc.release(); c->hDst.release(); c->hSegs.release(); c->hStatus.release(); c->hDense.release();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->profDecode) (void)hipFree(c->profDecode);
    if (c->profPipe) (void)hipFree(c->profPipe);
    if (c->profEncode) (void)hipFree(c->profEncode);
    for (int i = 0; i < 2; i++) { if (c->hpStage[i]) (void)hipHostFree(c->hpStage[i]); if (c->hpStageFree[i]) (void)hipEventDestroy(c->hpStageFree[i]); }
    if (c->hpH2D) (void)hipStreamDestroy(c->hpH2D);
    if (c->hpCompute) (void)hipStreamDestroy(c->hpCompute);
    if (c->hpD2H) (void)hipStreamDestroy(c->hpD2H);
    delete c;
}
extern "C" const char* zhip_kernel_name(int k)
{
    static const char* names[ZHIP_NTIMER] = {"zhip_decode_frames_kernel", "zhip_encode_frames_kernel", "zhip_decode_lit_kernel",
                                   "zhip_decode_seq_kernel", "zhip_decode_exec_kernel", "zhip_encode_match_kernel",
                                   "zhip_encode_entropy_kernel", "zhip_decode_huf_kernel", "zhip_encode_match_flat_kernel", "zhip_decode_pipeline_span"};
    return k >= 0 && k < ZHIP_NTIMER ? names[k] : "";
}
extern "C" int zhip_ctx_kernel_time(zhip_ctx* c, int direction, double* avgMs, uint64_t* launches)
{
    if (!c || direction < 0 || direction >= ZHIP_NTIMER) return ZHIP_ERR_UNSUPPORTED;
    c->timing = true;                           // from now on the launches of this context are bracketed by events
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < ZHIP_NTIMER; i++) drain_shared(c->timer[i]);
    KTimer& t = c->timer[direction];
    drain_timer(t);
    if (avgMs) *avgMs = t.launches ? t.totalMs / (double)t.launches : 0.0;
    if (launches) *launches = t.launches;
    t.totalMs = 0; t.launches = 0;
    return 0;
}

extern "C" int zhip_ctx_table_pick(zhip_ctx* c, float* ms3)
{
    if (!c) return 0;
    if (ms3) for (int k = 0; k < 3; k++) ms3[k] = c->e1fPickMs[k];
    return c->e1fPickKept;
}

static uint64_t dict_fingerprint(const void* p, size_t n, uint64_t salt)
{
    uint64_t h = 0xcbf29ce484222325ull ^ salt;
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    h ^= (uint64_t)n * 0x9E3779B97F4A7C15ull;
    return h ? h : 1;
}
extern "C" int zhip_ctx_set_ddict(zhip_ctx* c, const void* hostDict, size_t dictSize, int dictType)
{
    if (!c) return ZHIP_ERR_UNSUPPORTED;
    if (dictType != ZHIP_DICT_AUTO && dictType != ZHIP_DICT_RAWCONTENT && dictType != ZHIP_DICT_FULLDICT) { g_lastError = "invalid dictionary type"; return ZHIP_ERR_UNSUPPORTED; }
    if (hostDict && dictSize) {
        const uint64_t key = dict_fingerprint(hostDict, dictSize, 0x100u + (uint64_t)dictType);
        if (c->dictSize == dictSize && c->ddictKey == key) return 0;       // same dictionary as last time: tables are still resident
        c->ddictKey = key;
    } else c->ddictKey = 0;
    c->dictSize = 0; c->dictID = 0; c->dictContentOffset = 0; c->dictHasEntropy = false;
    if (!hostDict || !dictSize) return 0;
    // (the decode kernels pack an offset in 29 bits and keep the values from ZP_OF_LIMIT = 480 MiB up for other purposes: dictionary + one
    // frame's window must stay below that for "offset beyond everything decoded so far" to be decidable from the packed value --
    // zhip_decode_pipeline.hpp, K2. The limit is set at 256 MiB, half of it, leaving the other half to the frames' own windows (up to 2^27
    // here); the reference takes any size: larger dictionaries are refused loudly, INTEGRATION.md section 4 lists the limit)
    if (dictSize >= (256u << 20)) { g_lastError = "dictionary too large"; c->ddictKey = 0; return ZHIP_ERR_UNSUPPORTED; }
    // ZSTD_loadEntropy_intoDDict (zstd.c:42716): raw content when asked for, when shorter than 8 bytes or when the magic is absent --
    // unless a full dictionary was demanded, which is then "Dictionary is corrupted"
    const bool hasMagic = dictSize >= 8 && rd32((const uint8_t*)hostDict) == ZF_DICT_MAGIC;
    if (dictType == ZHIP_DICT_FULLDICT && !hasMagic) { c->ddictKey = 0; return -ZE_DICT_CORRUPTED; }
    if (c->dictBlob.reserve(dictSize + 64)) { c->ddictKey = 0; return g_reserveRc; }       // (+ 64: K3 reads whole 32-byte windows)
    if (c->dictEntropy.reserve(sizeof(ZhipDictEntropy))) { c->ddictKey = 0; return g_reserveRc; }
    HIP_TRY(hipMemcpy(c->dictBlob.p, hostDict, dictSize, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(c->dictEntropy.p, 0, sizeof(ZhipDictEntropy)));
    ZhipDictEntropy de; memset(&de, 0, sizeof de);
    if (hasMagic && dictType != ZHIP_DICT_RAWCONTENT) {
        hipLaunchKernelGGL(zhip_parse_dict_kernel, dim3(1), dim3(64), 0, 0, (const uint8_t*)c->dictBlob.p, (uint32_t)dictSize,
                           (ZhipDictEntropy*)c->dictEntropy.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(&de, c->dictEntropy.p, sizeof de, hipMemcpyDeviceToHost));
        if (de.status) { c->ddictKey = 0; return -de.status; }            // negative zstd error code: dictionary corrupted
        if (de.hufCount) {                                                // the decode pipeline's ready-made tables
            if (c->dictTables.reserve(sizeof(ZhipDictTables))) { c->ddictKey = 0; return g_reserveRc; }
            hipLaunchKernelGGL(zhip_dict_tables_kernel, dim3(1), dim3(64), 0, 0, (const ZhipDictEntropy*)c->dictEntropy.p, (ZhipDictTables*)c->dictTables.p);
            HIP_TRY(hipGetLastError());
            int32_t tst = 0;
            HIP_TRY(hipMemcpy(&tst, (const uint8_t*)c->dictTables.p + offsetof(ZhipDictTables, status), 4, hipMemcpyDeviceToHost));
            if (tst) { c->ddictKey = 0; return -ZE_DICT_CORRUPTED; }
        }
    }
    c->dictSize = (uint32_t)dictSize; c->dictID = de.dictID; c->dictContentOffset = de.contentOffset;
    c->dictHasEntropy = de.hufCount != 0;
    return 0;
}
extern "C" int zhip_ctx_set_dformat(zhip_ctx* c, int format, uint64_t maxWindowSize)
{
    if (!c || (format != ZHIP_FORMAT_ZSTD1 && format != ZHIP_FORMAT_ZSTD1_MAGICLESS)) { g_lastError = "invalid frame format"; return ZHIP_ERR_UNSUPPORTED; }
    c->dformat = format;
    c->maxWindowSize = maxWindowSize ? maxWindowSize : ((1ull << 27) + 1);
    return 0;
}
// bring-up probe: launches a 64-lane kernel exercising every wave primitive the codec uses; returns 0 when correct
extern "C" int zhip_selftest(void)
{
    uint32_t* d = nullptr; uint32_t h[64];
    HIP_TRY(hipMalloc((void**)&d, sizeof h));
    hipLaunchKernelGGL(zhip_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (uint32_t l = 0; l < 64; l++) if (h[l] != l * (l + 1) / 2 + 32 + 63 + 7) { g_lastError = "selftest mismatch"; return 1; }
    return 0;
}
extern "C" int zhip_ctx_set_cparams(zhip_ctx* c, const zhip_cparams* p)
{
    if (!c || !p) return ZHIP_ERR_UNSUPPORTED;
    if (p->dictType != ZHIP_DICT_AUTO && p->dictType != ZHIP_DICT_RAWCONTENT && p->dictType != ZHIP_DICT_FULLDICT) { g_lastError = "invalid dictionary type"; return ZHIP_ERR_UNSUPPORTED; }
    if (p->format != ZHIP_FORMAT_ZSTD1 && p->format != ZHIP_FORMAT_ZSTD1_MAGICLESS) { g_lastError = "invalid frame format"; return ZHIP_ERR_UNSUPPORTED; }
    // level + explicit parameters -> one row per source-size class; the kernels implement the fast and double-fast strategies
    ZeRows rows;
    zh_resolve_rows(&rows, p->level, &p->cp);
    bool any = false;
    for (int t = 0; t < 4; t++) {
        if (!zh_check_cparams(rows.r[t])) return -ZE_PARAM_OUTOFBOUND;        // what ZSTD_CCtx_setParametersUsingCCtxParams reports
        any |= rows.r[t][6] == 1 || rows.r[t][6] == 2;
    }
    if (!any) { g_lastError = "HIP backend compresses with the fast and double-fast strategies only (levels <= 3, negative levels; level 4 for inputs above 16 KiB; or explicit strategy / parameters that select them)"; return ZHIP_ERR_UNSUPPORTED; }
    // nothing of a dictionary shorter than 8 bytes is loaded (ZSTD_compress_insertDictionary, zstd.c:28167) -- unless a full dictionary
    // was demanded, which is then "Dictionary mismatch"; so is a blob without the magic. The ZSTD_CDict still exists and its
    // parameter row (chosen for a 513-byte source) is what the frames are compressed with, so such a blob is digested like any other
    const bool hasDict = p->dict && p->dictSize;
    const bool hasMagic = hasDict && p->dictSize >= 8 && rd32((const uint8_t*)p->dict) == ZF_DICT_MAGIC;
    // What the caller sees when the dictionary cannot be digested is libzstd's, quirk included: the reference hands the blob to
    // ZSTD_CCtx_loadDictionary_advanced (c-ext/compressor.c:34-37), which only stores it; the ZSTD_CDict is made at the first compression
    // (ZSTD_initLocalDict, zstd.c:24206) and ANY failure there -- dictionary_wrong, dictionary_corrupted -- comes back as a NULL CDict,
    // reported as memory_allocation (zstd.c:24232): "cannot compress: Allocation error : not enough memory".
    if (hasDict && p->dictType == ZHIP_DICT_FULLDICT && !hasMagic) return -ZE_MEMORY;
    const bool useDict = hasDict;
    if (useDict) {
        uint64_t salt = 0x200u + (uint64_t)p->dictType;
        for (int t = 0; t < 4; t++) for (int k = 0; k < 7; k++) salt = salt * 1000003u + (uint64_t)(uint32_t)rows.r[t][k];
        const uint64_t key = dict_fingerprint(p->dict, p->dictSize, salt);
        if (c->hasCDict && c->cdictKey == key) { c->cparams = *p; c->cparams.dict = nullptr; c->cparams.dictSize = 0; c->rows = rows; return 0; }
        c->cdictKey = key;
    }
    c->hasCDict = false;
    if (useDict) {
        // digest the dictionary on the device: parse its entropy section, build the encoding tables, index its content
        // (what ZSTD_createCDict does on the host in the reference, zstd.c:28490-28614)
        if (p->dictSize > 0x7FFFFFFFu) { g_lastError = "dictionary too large"; return ZHIP_ERR_UNSUPPORTED; }
        const size_t cells = (size_t)1 << ZE_CDICT_MAX_HLOG;
        if (c->cdictBlob.reserve(p->dictSize + 16) || c->cdictEntropy.reserve(sizeof(ZhipDictEntropy)) ||
            c->cdictDigest.reserve(sizeof(ZeCDict)) || c->cdictTables.reserve(3 * cells * sizeof(uint32_t))) return g_reserveRc;
        HIP_TRY(hipMemcpy(c->cdictBlob.p, p->dict, p->dictSize, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset((uint8_t*)c->cdictBlob.p + p->dictSize, 0, 16));
        HIP_TRY(hipMemset(c->cdictEntropy.p, 0, sizeof(ZhipDictEntropy)));
        HIP_TRY(hipMemset(c->cdictDigest.p, 0, sizeof(ZeCDict)));
        ZhipDictEntropy de; memset(&de, 0, sizeof de);
        if (hasMagic && p->dictType != ZHIP_DICT_RAWCONTENT) {
            hipLaunchKernelGGL(zhip_parse_dict_kernel, dim3(1), dim3(64), 0, 0, (const uint8_t*)c->cdictBlob.p, (uint32_t)p->dictSize,
                               (ZhipDictEntropy*)c->cdictEntropy.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(&de, c->cdictEntropy.p, sizeof de, hipMemcpyDeviceToHost));
            if (de.status) return -ZE_MEMORY;                   // (dictionary_corrupted inside ZSTD_createCDict: see above)
        }
        uint32_t* t = (uint32_t*)c->cdictTables.p;
        hipLaunchKernelGGL(zhip_build_cdict_kernel, dim3(1), dim3(64), 0, 0, (const uint8_t*)c->cdictBlob.p, (uint32_t)p->dictSize,
                           (const ZhipDictEntropy*)c->cdictEntropy.p, rows, (ZeCDict*)c->cdictDigest.p, t, t + cells, t + 2 * cells);
        HIP_TRY(hipGetLastError());
        ZeCDict cd;
        HIP_TRY(hipMemcpy(&cd, c->cdictDigest.p, sizeof cd, hipMemcpyDeviceToHost));
        if (cd.status == ZE_PARAM_UNSUPPORTED) { g_lastError = "dictionary / level combination outside the fast / double-fast attached-dictionary paths of the HIP backend"; return ZHIP_ERR_UNSUPPORTED; }
        if (cd.status) return cd.status == ZE_DICT_CORRUPTED || cd.status == ZE_DICT_WRONG ? -ZE_MEMORY : -cd.status;
        c->hasCDict = true; c->cdictContentOffset = de.hufCount ? de.contentOffset : 0u;
        c->cdictAttachMax = cd.strat == 1 ? ZE_DICT_ATTACH_MAX_FAST : ZE_DICT_ATTACH_MAX;
        c->cdictHlog = cd.hlog; c->cdictClog = cd.clog; c->cdictStrat = cd.strat; c->cdictContentSize = cd.contentSize;
    }
    c->cparams = *p; c->cparams.dict = nullptr; c->cparams.dictSize = 0; c->rows = rows;
    return 0;
}

// ------------------------------------------------------------------------------------------ device-resident decode
extern "C" int zhip_decompress_batch_device(zhip_ctx* c, const void* d_src, const zhip_segment* d_srcSegs, size_t n,
                                            void* d_dst, const zhip_segment* d_dstSegs, uint64_t* d_outSizes,
                                            int32_t* d_status, void* streamv)
{
    if (!c) return ZHIP_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFu) { g_lastError = "too many frames in one launch"; return ZHIP_ERR_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)streamv;
    size_t maxBlocks = (size_t)c->numCU * (size_t)c->decBlocksPerCU;
    uint32_t grid = (uint32_t)(n < maxBlocks ? n : maxBlocks);
    if (c->scratch.reserve((size_t)grid * ZHIP_LIT_STRIDE)) return g_reserveRc;
    if (c->counter.reserve(64)) return g_reserveRc;
    HIP_TRY(hipMemsetAsync(c->counter.p, 0, 4, stream));
    const uint32_t* d_fallbackList = nullptr; const uint32_t* d_fallbackCount = nullptr;
    const bool usePipeline = true;
    if (usePipeline) {
        // phase-split fast path for single-block, dictionary-less frames (zhip_decode_pipeline.hpp); everything it declines
        // lands in the fallback list consumed by the generic kernel below.
        // Chunks of frames flow through K1 -> K2 -> K3 on ZHIP_NSLOT internal streams (slot = chunk % NSLOT, each slot has its own
        // arenas and counters), so that different chunks' kernels overlap on the GPU: each phase is latency-bound with idle issue
        // slots, and their LDS footprints differ, which is exactly when co-residency pays.
        const size_t chunkMax = c->knob.dchunk; int slotMax = c->knob.nslot;      // measured on MI355X (profiles/README.md, r01c / r02f / r02zl): 65 536 frames per chunk, 2 slots
        // Frames of several blocks (the caller's size hint says so: the host API sets it from the items it sees): the several-block mode --
        // the arenas' slots are per BLOCK, a chunk is as many frames as fit `chunkMax` slots at the estimate below (libzstd cuts a block of
        // 128 KiB where the data changes; frames with more blocks than their share still work while the chunk has slots left, then
        // they are the generic kernel's)
        const uint64_t sizeHint = c->dstMaxHint ? c->dstMaxHint : c->itemHint;
        // (small frames -- the caller says so -- are short kernels with launch gaps between them: a third chunk slot fills them. 262 144 x 4 KiB with the
        // shared dictionary: 123 -> 134 GB/s, r05q; frames of 128 KiB: the kernels are long, two slots overlap little as it is)
        if (!c->knob.nslotSet && sizeHint && sizeHint <= 16384 && slotMax < 3) slotMax = 3;
        const bool mb = sizeHint > ZF_BLOCK_MAX && sizeHint <= 0x7FFFFFFFull;     // (larger frames are the generic kernel's anyway)
        // K1b beside K2 on a side stream (below) -- not for batches of small frames: their kernels are short, the three chunk slots overlap them already, and the
        // side streams only added queues (262 144 x 4 KiB with the dictionary: 199 -> 187 GB/s, r06p)
        const bool side = ZHIP_SIDE && !c->hostPipe && !(sizeHint && sizeHint <= 16384);
        // (with the side stream a batch of several chunks runs them one after the other on ONE slot stream: two slots' kernels side by side mix K3 with the next chunk's
        // K2 -- each slows the other, section 4.1 of DESIGN.md -- and the side stream has taken the tail the second slot used to fill. 131 072 x 128 KiB: 360 GB/s on two
        // slots with or without the side stream, 365-366 on one slot with it, and half the scratch (r06s). A single 128 KiB frame: 3.6 -> 2.4 ms, its two chains run together)
        if (side) slotMax = 1;
        // (two slots per 128 KiB decide how many frames make a chunk -- the slots are a pool, a frame may take more than its share; a chunk of
        // FEW frames has no pool to lean on and gets four: 64 x 128 KiB of changing data in one frame came as 235 blocks, r03x)
        // (the host-buffer API knows every frame's size and says how many slots the batch should need in all -- a batch of mostly small frames
        // with a few large ones is then not cut into chunks sized for the large ones: chunks are made for half the pool, the pool is twice
        // what the chunk is expected to use)
        const size_t perFrame = mb ? (size_t)(2 * ((sizeHint + ZF_BLOCK_MAX - 1) / ZF_BLOCK_MAX) + 2) : 1;
        const size_t slotsHint = mb ? c->dstSlotsHint : 0;
        const size_t chunkFrames = !mb ? chunkMax : slotsHint ? (size_t)((double)n * (double)(chunkMax / 2) / (double)slotsHint) + 1 : (chunkMax / perFrame ? chunkMax / perFrame : 1);
        const size_t chunk = n < chunkFrames ? n : chunkFrames;
        auto slotsFor = [&](size_t frames) {
            if (!mb) return frames;
            if (slotsHint) { const size_t want = 2 * (size_t)((double)frames * (double)slotsHint / (double)n + 1.0) + 64; return want < frames ? frames : want; }
            const size_t lo = frames * perFrame, hi = 2 * lo < chunkMax ? 2 * lo : chunkMax; return lo > hi ? lo : hi; };
        const size_t slots = slotsFor(chunk);                                           // item slots per chunk (== frames without the mode)
        if (slots > 0x7FFFFFFFu) { g_lastError = "frame too large for the block arenas"; return ZHIP_ERR_UNSUPPORTED; }
        const size_t nChunks = (n + chunk - 1) / chunk;
        const int nslot = (int)(nChunks < (size_t)slotMax ? nChunks : (size_t)slotMax);
        // ONE compact arena holds literals and sequences (a "frame" below is an ITEM -- a block -- in the several-block mode): K1 claims what a
        // frame's literals need, K2 what a group's sequences need, from a per-chunk budget (ZhipPipeArgs.bases). A block's literals and sequences
        // trade off -- every sequence costs at least three bytes of output -- so the budget is common: 160 KiB per frame on average (the ratio-3 bench
        // corpus needs 45 KiB + 76 KiB, Huffman-only data 128 KiB + nothing, 4-symbol noise 3 + 150 KiB; the worst case, a match every three bytes, 350 KiB):
        // 10 GiB per 65 536-frame chunk instead of the 8 + 22 GiB of fixed slots, with up to 1 024 frames' worth of worst case as the floor so that small
        // batches never run out. What does not fit goes to the generic kernel (correct, slower).
        const size_t floorFrames = slots < 1024 ? slots : 1024;
        // (frames the caller says are SMALL cannot need more than their own worst case -- literals of the frame's size + 256, a sequence per three bytes in
        // rooms of the group's longest + 8 --: 4 x the size + 1 KiB; 4 KiB documents get 17 KiB each instead of 160)
        const size_t smallWorst = !mb && sizeHint && sizeHint < ZF_BLOCK_MAX ? 4 * (size_t)sizeHint + 1024 : (size_t)0;
        const size_t roomPerFrame = smallWorst && smallWorst < (160u << 10) ? smallWorst : (size_t)(160u << 10);
        size_t arenaBudget16 = slots * (roomPerFrame / 16);
        const size_t worstFrame = smallWorst ? smallWorst : (size_t)(ZP_SEQ_STRIDE + ZP_LIT_STRIDE);
        if (arenaBudget16 < floorFrames * (worstFrame / 16)) arenaBudget16 = floorFrames * (worstFrame / 16);
        if (arenaBudget16 > 0xFFFFFFF0u) arenaBudget16 = 0xFFFFFFF0u;
        const size_t arenaBytes = arenaBudget16 * 16 + 512;
        if (c->pipeBases.reserve(nslot * slots * 2 * sizeof(uint32_t))) return g_reserveRc;
        if (c->pipeMeta.reserve(nslot * slots * sizeof(ZdMeta)) || c->pipeLit.reserve(nslot * arenaBytes + ZP_LIT_FRONT) || c->pipeCounters.reserve((8 + (size_t)ZHIP_NSLOT * ZP_CNT_WORDS) * 4) || c->pipeFallback.reserve(n * 4 + 16) ||
            c->pipeFse.reserve(nslot * slots * ZP_FSE_CELLS * sizeof(uint16_t)) || c->pipeOrder.reserve(nslot * slots * sizeof(uint32_t)) ||
            c->pipeHuf.reserve(nslot * slots * ZP_HUF_CELLS * sizeof(uint16_t)) || c->pipeOrderLit.reserve(nslot * slots * sizeof(uint32_t))) return g_reserveRc;
        // K0's records (one per frame; not in the several-block mode, whose K1 walks a frame's blocks in order, nor for dictionary batches, whose lane pass finishes what has no table of its own)
        // (... nor for small batches: K0 is a lane-serial walk of ~0.1 ms whatever the batch, which pays from ~6 000 frames on -- 2 048 frames 3.53 -> 3.62 ms with it, 8 192
        // 4.66 -> 4.62, 16 384 7.17 -> 6.96, 32 768 12.5 -> 12.0, 65 536 23.4 -> 22.4; `profiles/r06z2_k0_by_batch_size.txt`)
        const bool pre = ZHIP_K0 && !mb && !c->dictHasEntropy && chunk >= c->knob.k0Min;
        if (pre && c->pipePre.reserve(nslot * slots * sizeof(ZpPre))) return g_reserveRc;
        if (mb && (c->pipeItemFrame.reserve(nslot * slots * sizeof(uint32_t)) || c->pipeItemReps.reserve(nslot * slots * 4 * sizeof(uint32_t)) ||
                   c->pipeFrameRecs.reserve(nslot * chunk * sizeof(ZpFrameRec)))) return g_reserveRc;
        for (int sidx = 0; sidx < nslot; sidx++) if (!c->slotStream[sidx]) HIP_TRY(hipStreamCreateWithFlags(&c->slotStream[sidx], hipStreamNonBlocking));
        HIP_TRY(hipMemsetAsync(c->pipeCounters.p, 0, (8 + (size_t)ZHIP_NSLOT * ZP_CNT_WORDS) * 4, stream));
        hipEvent_t evStart; HIP_TRY(hipEventCreateWithFlags(&evStart, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(evStart, stream));
        for (int sidx = 0; sidx < nslot; sidx++) HIP_TRY(hipStreamWaitEvent(c->slotStream[sidx], evStart, 0));
        (void)hipEventDestroy(evStart);
        ZhipPipeArgs pa; memset(&pa, 0, sizeof pa);
        pa.src = (const uint8_t*)d_src; pa.srcSegs = (const uint64_t*)d_srcSegs; pa.dst = (uint8_t*)d_dst; pa.dstSegs = (const uint64_t*)d_dstSegs;
        pa.outSizes = d_outSizes; pa.status = d_status; pa.fallbackList = (uint32_t*)c->pipeFallback.p;
        pa.maxWindowSize = c->maxWindowSize; pa.magicless = c->dformat == ZHIP_FORMAT_ZSTD1_MAGICLESS;
        if (c->dictSize) {                                  // dictionary frames take the pipeline too (r02v; the generic kernel rebuilt the dictionary's tables per frame)
            pa.dictID = c->dictID;
            pa.dictContent = (const uint8_t*)c->dictBlob.p + c->dictContentOffset;
            pa.dictContentSize = c->dictSize - c->dictContentOffset;
            pa.dictEntropy = c->dictHasEntropy ? (const ZhipDictEntropy*)c->dictEntropy.p : nullptr;
            pa.dictTables = c->dictHasEntropy ? (const ZhipDictTables*)c->dictTables.p : nullptr;
        }
        if (c->knob.prof) {
            if (!c->profPipe && (c->dictSize || mb))
                fprintf(stderr, "[zhip-prof] K3's phase timers exist in the dictionary-less single-block kernel only: with a dictionary or frames of several blocks they read zero\n");
            if (!c->profPipe) HIP_TRY(hipMalloc((void**)&c->profPipe, 32 * 8));
            HIP_TRY(hipMemsetAsync(c->profPipe, 0, 32 * 8, stream));
            pa.prof = c->profPipe;
        }
        uint32_t* counters = (uint32_t*)c->pipeCounters.p;        // [0] = fallback length, then 8 words per slot
        pa.fallbackCount = counters;
        size_t ci = 0;
        for (size_t first = 0; first < n; first += chunk, ci++) {
            const int sidx = (int)(ci % nslot);
            hipStream_t ss = c->slotStream[sidx];
            const size_t cnt = n - first < chunk ? n - first : chunk;
            pa.first = (uint32_t)first; pa.count = (uint32_t)cnt;
            pa.meta = (ZdMeta*)c->pipeMeta.p + (size_t)sidx * slots;
            pa.litArena = (uint8_t*)c->pipeLit.p + ZP_LIT_FRONT + (size_t)sidx * arenaBytes;     // (ZP_LIT_FRONT = 256 bytes in front: K2's parked / first stores of a room at 0 land there)
            pa.seqArena = (uint64_t*)pa.litArena;
            pa.bases = (uint32_t*)c->pipeBases.p + (size_t)sidx * slots * 2; pa.arenaBudget16 = (uint32_t)arenaBudget16;
            pa.fseTables = (uint16_t*)c->pipeFse.p + (size_t)sidx * slots * ZP_FSE_CELLS;
            pa.order = (uint32_t*)c->pipeOrder.p + (size_t)sidx * slots;
            pa.hufTables = (uint16_t*)c->pipeHuf.p + (size_t)sidx * slots * ZP_HUF_CELLS;
            pa.orderLit = (uint32_t*)c->pipeOrderLit.p + (size_t)sidx * slots;
            if (mb) {
                pa.itemCap = (uint32_t)slotsFor(cnt);
                pa.itemFrame = (uint32_t*)c->pipeItemFrame.p + (size_t)sidx * slots;
                pa.itemReps = (uint32_t*)c->pipeItemReps.p + (size_t)sidx * slots * 4;
                pa.frameRecs = (ZpFrameRec*)c->pipeFrameRecs.p + (size_t)sidx * chunk;
            }
            const size_t items = mb ? (size_t)pa.itemCap : cnt;                         // K1b / K2 / KB work items (an upper bound in the several-block mode)
            pa.counters = counters + 8 + ZP_CNT_WORDS * sidx;
            if (ci >= (size_t)nslot) HIP_TRY(hipMemsetAsync(pa.counters, 0, ZP_CNT_WORDS * 4, ss));
            size_t g1m = (size_t)c->numCU * c->k1PerCU, g3m = (size_t)c->numCU * c->k3PerCU;
            if (c->knob.k3PerCU) g3m = (size_t)c->numCU * (size_t)c->knob.k3PerCU;
            if (c->knob.k1PerCU) g1m = (size_t)c->numCU * (size_t)c->knob.k1PerCU;
            // K2 and K1b are sized by LDS: as many one-wave workgroups per CU as their table sets fit (K2: 15 frames per wave -> 4)
            const size_t w2 = (items + ZQ_FRAMES - 1) / ZQ_FRAMES, g2m = (size_t)c->numCU * (ZHIP_LDS_BYTES / sizeof(ZpSeqQLDS));
            // dictionary batches: a lane-per-frame pass first (zhip_decode_lit_lanes_kernel: frames whose tables are all the dictionary's are nothing but header
            // arithmetic), K1 then only over the frames that pass listed
            pa.k1Lanes = !mb && c->dictHasEntropy ? 1u : 0u;
            pa.ckLater = 1;
            pa.pre = pre ? (ZpPre*)c->pipePre.p + (size_t)sidx * slots : nullptr;
            const size_t tasks1 = cnt;
            const uint32_t g1 = (uint32_t)(tasks1 < g1m ? tasks1 : g1m), g2 = (uint32_t)(w2 < g2m ? w2 : g2m), g3 = (uint32_t)(cnt < g3m ? cnt : g3m);
            const size_t wh = (items + ZP_HUF_FRAMES - 1) / ZP_HUF_FRAMES, ghm = (size_t)c->numCU * (ZHIP_LDS_BYTES / sizeof(ZpHufKernelLDS));
            const uint32_t gh = (uint32_t)(wh < ghm ? wh : ghm);
            const bool tm = c->timing;
            hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}, evh = nullptr, evh2 = nullptr;
            if (tm) {
                for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&ev[i]));
                HIP_TRY(hipEventCreate(&evh)); HIP_TRY(hipEventCreate(&evh2));
                HIP_TRY(hipEventRecord(ev[0], ss));
            }
            if (mb) hipLaunchKernelGGL(zhip_decode_lit_mb_kernel, dim3(g1), dim3(64), 0, ss, pa);
            else {
                if (pa.pre) { const size_t w = (cnt + 63) / 64, gm = (size_t)c->numCU * 7; hipLaunchKernelGGL(zhip_decode_pre_kernel, dim3((uint32_t)(w < gm ? w : gm)), dim3(64), 0, ss, pa); }      // (7 waves of 21.3 KiB of LDS per CU)
                if (pa.k1Lanes) { const size_t w = (cnt + 63) / 64; hipLaunchKernelGGL(zhip_decode_lit_lanes_kernel, dim3((uint32_t)(w < g1m ? w : g1m)), dim3(64), 0, ss, pa); }
                hipLaunchKernelGGL(zhip_decode_lit_kernel, dim3(g1), dim3(64), 0, ss, pa);
            }
            hipLaunchKernelGGL(zhip_decode_bin_kernel, dim3(2 * (items < 4096 ? 1u : 64u)), dim3(64), 0, ss, pa);      // tiny; timed with K1
            hipEvent_t evS0 = nullptr, evS1 = nullptr;
            if (side) {
                // K1b and K2 need nothing of each other: both follow K1 and the bins, K3 follows both. K2's 15-frame groups are long (~1.9 ms at 128 KiB) and a wave gets 4.27 of
                // them at 65 536 frames, so K2 ends with three quarters of its waves gone for a group's time, and K1b then starts on an empty chip. K1b goes to a side stream BEHIND
                // K2's launch instead: K2's one-wave workgroups are all resident at once, K1b's land where K2's have drained (two K2 waves' LDS make room for one of K1b's).
                // 65 536 x 128 KiB: 24.35 -> 23.3 ms (K1b's 3.35 ms cost 2.4 beyond K2's end), frames of several blocks 150 -> 158 GB/s (r06p). The other order (K1b first)
                // LOSES, 26.2 ms: K1b's short groups keep handing LDS to K2 piecemeal and the mix holds three waves per CU where K2 alone holds four.
                if (!c->sideStream[sidx]) HIP_TRY(hipStreamCreateWithFlags(&c->sideStream[sidx], hipStreamNonBlocking));
                hipStream_t sd = c->sideStream[sidx];
                hipEvent_t evBin, evSide; HIP_TRY(hipEventCreateWithFlags(&evBin, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&evSide, hipEventDisableTiming));
                if (tm) { HIP_TRY(hipEventRecord(evh, ss)); HIP_TRY(hipEventRecord(evh2, ss)); HIP_TRY(hipEventCreate(&evS0)); HIP_TRY(hipEventCreate(&evS1)); }
                HIP_TRY(hipEventRecord(evBin, ss));
                HIP_TRY(hipStreamWaitEvent(sd, evBin, 0));
                if (mb) hipLaunchKernelGGL(zhip_decode_seq_mb_kernel, dim3(g2), dim3(64), 0, ss, pa);
                else hipLaunchKernelGGL(zhip_decode_seq_kernel, dim3(g2), dim3(64), 0, ss, pa);
                if (tm) { HIP_TRY(hipEventRecord(ev[1], ss)); HIP_TRY(hipEventRecord(evS0, ss)); }       // (two events at K2's end: one closes K2's pair, one opens K1b's)
                hipLaunchKernelGGL(zhip_decode_huf_kernel, dim3(gh), dim3(64), 0, sd, pa);
                if (tm) HIP_TRY(hipEventRecord(evS1, sd));
                HIP_TRY(hipEventRecord(evSide, sd));
                HIP_TRY(hipStreamWaitEvent(ss, evSide, 0));
                (void)hipEventDestroy(evBin); (void)hipEventDestroy(evSide);
                if (tm) HIP_TRY(hipEventRecord(ev[2], ss));
            } else {
                if (tm) { HIP_TRY(hipEventRecord(evh, ss)); HIP_TRY(hipEventRecord(evh2, ss)); }
                hipLaunchKernelGGL(zhip_decode_huf_kernel, dim3(gh), dim3(64), 0, ss, pa);
                if (tm) HIP_TRY(hipEventRecord(ev[1], ss));
                if (mb) hipLaunchKernelGGL(zhip_decode_seq_mb_kernel, dim3(g2), dim3(64), 0, ss, pa);
                else hipLaunchKernelGGL(zhip_decode_seq_kernel, dim3(g2), dim3(64), 0, ss, pa);
                if (tm) HIP_TRY(hipEventRecord(ev[2], ss));
            }
            if (mb && pa.dictContent) hipLaunchKernelGGL(zhip_decode_exec_mb_dict_kernel, dim3(g3), dim3(64), 0, ss, pa);
            else if (mb) hipLaunchKernelGGL(zhip_decode_exec_mb_kernel, dim3(g3), dim3(64), 0, ss, pa);
            else if (pa.dictContent) hipLaunchKernelGGL(zhip_decode_exec_dict_kernel, dim3(g3), dim3(64), 0, ss, pa);
            else if (pa.prof) hipLaunchKernelGGL(zhip_decode_exec_prof_kernel, dim3(g3), dim3(64), 0, ss, pa);      // (ZHIP_PROF with a dictionary or frames of several blocks: K3's timers read zero -- said once at context creation)
            else hipLaunchKernelGGL(zhip_decode_exec_kernel, dim3(g3), dim3(64), 0, ss, pa);
            { const size_t w = (cnt + 63) / 64, gm = (size_t)c->numCU * 8; hipLaunchKernelGGL(zhip_decode_check_kernel, dim3((uint32_t)(w < gm ? w : gm)), dim3(64), 0, ss, pa); }      // KX (returns at once when no frame carries a checksum; timed with K3)
            if (tm) HIP_TRY(hipEventRecord(ev[3], ss));
            HIP_TRY(hipGetLastError());
            if (tm) {
                // consecutive events bracket one kernel each (same stream, nothing in between). Ownership: K1's timer owns
                // (ev[0], evh), K1b's (evh2, ev[1]), K3's (ev[2], ev[3]); K2's pair (ev[1], ev[2]) is borrowed and always drained
                // before any destroy.
                c->timer[2].pending.emplace_back(ev[0], evh);          // K1 (+ the two bin waves)
                if (side) {
                    // K2's pair is its own now; K1b's is what it costs the step: from K2's END (on the main stream) to its own end on the side stream -- its first part
                    // runs inside K2's tail --, so that the kernels' times still add up to the step (a K1b that ends before K2 counts zero)
                    c->timer[3].pending.emplace_back(evh2, ev[1]);
                    c->timer[7].pending.emplace_back(evS0, evS1);
                } else {
                    c->timer[7].pending.emplace_back(evh2, ev[1]);         // K1b
                    c->timer[3].shared.emplace_back(ev[1], ev[2]);
                }
                c->timer[9].shared.emplace_back(ev[0], ev[3]);             // the chunk's pipeline, K1's start to K3's end: what the step costs now that two of its kernels overlap
                c->timer[4].pending.emplace_back(ev[2], ev[3]);
            }
        }
        if (pa.prof) {
            HIP_TRY(hipDeviceSynchronize());
            unsigned long long h[32];
            HIP_TRY(hipMemcpy(h, pa.prof, sizeof h, hipMemcpyDeviceToHost));
            static const char* nm[10] = {"header", "huf-table", "huf-decode", "seq-header+fse", "K3 load+scan", "-", "K3 literal/far fetch", "K3 near rounds", "K3 flush", "tail"};
            for (int k = 0; k < 2; k++) {
                unsigned long long tot = 0; for (int q = 0; q < 10; q++) tot += h[16 * k + q];
                fprintf(stderr, "[zhip-prof] %s: %.0f wave-cycles per frame\n", k ? "K3" : "K1", (double)tot / (double)n);
                for (int q = 0; q < 10; q++) if (h[16 * k + q]) fprintf(stderr, "[zhip-prof]    %-22s %6.2f%% %10.0f cyc/frame\n", nm[q], 100.0 * h[16 * k + q] / (tot ? tot : 1), (double)h[16 * k + q] / (double)n);
            }
        }
        for (int sidx = 0; sidx < nslot; sidx++) {                 // the caller's stream continues after every slot has drained
            hipEvent_t evEnd; HIP_TRY(hipEventCreateWithFlags(&evEnd, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(evEnd, c->slotStream[sidx]));
            HIP_TRY(hipStreamWaitEvent(stream, evEnd, 0));
            (void)hipEventDestroy(evEnd);
        }
        d_fallbackList = (const uint32_t*)c->pipeFallback.p; d_fallbackCount = (const uint32_t*)c->pipeCounters.p;
        // the generic kernel only sees the (usually empty) fallback list: a small grid is enough
        // (unless the caller says its frames exceed one block: then the list is the whole batch)
        if (grid > 1024 && (mb || !(sizeHint > ZF_BLOCK_MAX))) grid = 1024;
    }
    ZhipDecodeArgs a; memset(&a, 0, sizeof a);
    a.src = (const uint8_t*)d_src; a.srcSegs = (const uint64_t*)d_srcSegs; a.dst = (uint8_t*)d_dst;
    a.dstSegs = (const uint64_t*)d_dstSegs; a.outSizes = d_outSizes; a.status = d_status;
    a.scratch = (uint8_t*)c->scratch.p; a.counter = (uint32_t*)c->counter.p; a.n = (uint32_t)n;
    a.maxWindowSize = c->maxWindowSize; a.magicless = c->dformat == ZHIP_FORMAT_ZSTD1_MAGICLESS;
    a.frameList = d_fallbackList; a.listCount = d_fallbackCount;
    if (c->dictSize) {
        a.dictID = c->dictID;
        a.dictContent = (const uint8_t*)c->dictBlob.p + c->dictContentOffset;
        a.dictContentSize = c->dictSize - c->dictContentOffset;
        a.dictEntropy = c->dictHasEntropy ? (const ZhipDictEntropy*)c->dictEntropy.p : nullptr;
    }
    const bool prof = c->knob.prof;
    if (prof) {
        if (!c->profDecode) HIP_TRY(hipMalloc((void**)&c->profDecode, ZP_N * 8));
        HIP_TRY(hipMemsetAsync(c->profDecode, 0, ZP_N * 8, stream));
        a.prof = c->profDecode;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    hipLaunchKernelGGL(zhip_decode_frames_kernel, dim3(grid), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    if (c->timing) HIP_TRY(hipEventRecord(e1, stream));
    if (c->timing) c->timer[0].pending.emplace_back(e0, e1);
    if (prof) {
        unsigned long long h[ZP_N];
        HIP_TRY(hipMemcpy(h, c->profDecode, sizeof h, hipMemcpyDeviceToHost));
        static const char* names[ZP_N] = {"header/misc", "huf-table", "huf-decode", "seq-tables", "stage", "seq-decode(lane0)", "exec1(global->lds)", "exec2(lds rounds)", "flush", "raw/lastlit"};
        unsigned long long tot = 0; for (int i = 0; i < ZP_N; i++) tot += h[i];
        fprintf(stderr, "[zhip-prof] grid=%u (CUs %d x %d blocks) frames=%u wave-cycles total=%.3e (%.0f per frame)\n", grid, c->numCU, c->decBlocksPerCU, a.n, (double)tot, (double)tot / a.n);
        for (int i = 0; i < ZP_N; i++) fprintf(stderr, "[zhip-prof]   %-22s %6.2f%%  %10.0f cyc/frame\n", names[i], 100.0 * h[i] / (tot ? tot : 1), (double)h[i] / a.n);
    }
    if (c->timer[0].pending.size() > 1024) { HIP_TRY(hipStreamSynchronize(stream)); for (int i = 0; i < ZHIP_NTIMER; i++) drain_shared(c->timer[i]); for (int i = 0; i < ZHIP_NTIMER; i++) drain_timer(c->timer[i]); }
    return 0;
}
// the flat match kernel at `probes` per trip over `cnt` sources
static void launch_flat(int probes, size_t cnt, hipStream_t stream, const ZhipEncodeArgs& a)
{
    const dim3 g((uint32_t)((cnt + ZE_FLAT_LANES - 1) / ZE_FLAT_LANES)), b(64);
    if (probes == 4) hipLaunchKernelGGL(zhip_encode_match_flat4_kernel, g, b, 0, stream, a);
    else if (probes == 3) hipLaunchKernelGGL(zhip_encode_match_flat3_kernel, g, b, 0, stream, a);
    else hipLaunchKernelGGL(zhip_encode_match_flat_kernel, g, b, 0, stream, a);
}
extern "C" int zhip_compress_batch_device(zhip_ctx* c, const void* d_src, const zhip_segment* d_srcSegs, size_t n,
                                          void* d_dst, const zhip_segment* d_dstSegs, uint64_t* d_outSizes,
                                          int32_t* d_status, void* streamv)
{
    if (!c) return ZHIP_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFu) { g_lastError = "too many frames in one launch"; return ZHIP_ERR_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)streamv;
    size_t maxBlocks = (size_t)c->numCU * (size_t)c->encBlocksPerCU;
    uint32_t grid = (uint32_t)(n < maxBlocks ? n : maxBlocks);
    if (c->counter.reserve(64 * 8)) return g_reserveRc;
    uint8_t* const cbase = (uint8_t*)c->counter.p;
    HIP_TRY(hipMemsetAsync(cbase + 8, 0, 4, stream));
    ZhipEncodeArgs a; memset(&a, 0, sizeof a);
    a.src = (const uint8_t*)d_src; a.srcSegs = (const uint64_t*)d_srcSegs; a.dst = (uint8_t*)d_dst;
    a.dstSegs = (const uint64_t*)d_dstSegs; a.outSizes = d_outSizes; a.status = d_status;
    a.workspace = (uint8_t*)c->encWorkspace.p; a.counter = (uint32_t*)(cbase + 8); a.n = (uint32_t)n;
    a.level = c->cparams.level == 0 ? 3 : c->cparams.level;
    a.rows = c->rows; a.magicless = c->cparams.format == ZHIP_FORMAT_ZSTD1_MAGICLESS;
    a.contentSizeFlag = c->cparams.contentSizeFlag != 0; a.checksumFlag = c->cparams.checksumFlag != 0; a.dictIDFlag = c->cparams.dictIDFlag != 0;
    if (c->hasCDict) {
        const size_t cells = (size_t)1 << ZE_CDICT_MAX_HLOG;
        a.cdict = (const ZeCDict*)c->cdictDigest.p;
        a.cdictContent = (const uint8_t*)c->cdictBlob.p + c->cdictContentOffset;
        a.cdictHashLong = (const uint32_t*)c->cdictTables.p;
        a.cdictHashSmall = (const uint32_t*)c->cdictTables.p + cells;
    }
    {
        // two kernels: E1 searches with one LANE per frame (frames in flight hide the probe latency), E2 entropy-codes with one
        // wave per frame. Frames are processed in chunks so that the per-frame sequence/literal arena stays bounded.
        // table bytes of the largest one-block source of the two size classes these kernels serve (<= 128 KiB, <= 16 KiB), after the
        // per-source adjustment ze_get_cparams makes (windowLog <= 17 / 14, hashLog <= windowLog + 1, chainLog <= windowLog)
        bool anyDfast = false; uint32_t stride = 0;
        for (int t = 2; t < 4; t++) {
            const int32_t* r = a.rows.r[t];
            if (r[6] != 1 && r[6] != 2) continue;
            const int w = r[0] < (t == 2 ? 17 : 14) ? r[0] : (t == 2 ? 17 : 14);
            const int h = r[2] > w + 1 ? w + 1 : r[2], cl = r[1] > w ? w : r[1];
            const uint32_t bytes = (4u << h) + (r[6] == 2 ? (4u << cl) : 0u);
            if (bytes > stride) stride = bytes;
            anyDfast |= r[6] == 2;
        }
        // (sources of several blocks in the flat kernel -- `mbc` below: the rows of the larger size classes count too, at the hint's size)
        const size_t sizeHint = c->srcMaxHint ? c->srcMaxHint : c->itemHint;
        // (worth it from ~8 000 sources on. Either kernel's time is a source's serial chain of probes -- ~0.2 s per 128 KiB of source at these
        // batch sizes -- times the ROUNDS it needs: the generic kernel holds ~4 096 searches at a time (one per wave), the flat one a whole chunk
        // (64 per wave, whose every trip waits for the slowest of 64 requests and runs every lane's branch: ~10 % slower per round at 256 KiB,
        // 50 % at 512 KiB+). r03u / r03v, flat against generic: 2 048 x 1 MiB 2.5 s / 1.04 s, 4 096 x 512 KiB 1.31 / 0.85, 4 096 x 256 KiB
        // 0.44 / 0.42, 8 192 x 256 KiB 0.47 / 0.68, 16 384 x 256 KiB 0.51 / 1.24, 8 192 x 1 MiB 3.3 / ~2.1: the longer the sources the more
        // rounds of the generic kernel it takes to lose, hence the threshold grows with the size hint -- ZHIP_MBC_MIN sources per 256 KiB of it.
        // Round 6's kernels, flat / generic (r06zzi): 3 072 x 256 KiB 340 / 325 ms, 4 096 x 256 KiB 351 / 420, 6 144 x 384 KiB 810 / 846, 4 096 x 512 KiB 1 075 / 860,
        // 8 192 x 512 KiB 1 191 / 1 417, 12 288 x 512 KiB 1 639 / 1 965, 2 048 x 1 MiB 1 950 / 1 034: the flat search wins from 4 096 per 256 KiB on, was 8 192)
        const bool mbcWanted = anyDfast && !c->hasCDict && n >= c->knob.mbcMin * ((sizeHint + (256u << 10) - 1) / (256u << 10)) && sizeHint > ZF_BLOCK_MAX && sizeHint < ((size_t)1 << ZE_MB_POS_BITS) - 8;
        if (mbcWanted) for (int t = 0; t < 2; t++) {
            const int32_t* r = a.rows.r[t];
            if (r[6] != 2) continue;
            int w = 17; while (((size_t)1 << w) < sizeHint) w++;
            if (w > r[0]) w = r[0];
            const int h = r[2] > w + 1 ? w + 1 : r[2], cl = r[1] > w ? w : r[1];
            const uint32_t bytes = (4u << h) + (4u << cl);
            if (bytes > stride && bytes <= (12u << 17)) stride = bytes;
        }
        if (stride < (4u << 10)) stride = 4u << 10;
        if (stride > (12u << 17)) stride = 12u << 17;                  // larger tables: the frame is refused loudly by the match kernels
        a.arenaStride = (uint32_t)ZE_ARENA_STRIDE; a.arenaLit = ZE_ARENA_LIT;
        if (c->hasCDict) {
            // dictionary batches: every source is below the attach cutoff, so the per-lane tables are the dictionary row's shrunk to that
            // size (ze_dict_cparams) and a frame's sequences + literals fit a slot of the cutoff's size -- a tenth of the 128 KiB shapes,
            // which is what lets a whole 262 144-document batch be one chunk
            // (round 4: where the caller says how large its sources are -- the host API knows, a device-API caller can tell with
            // zhip_ctx_set_size_hint -- the slots are sized for THAT: 4 KiB documents need 48 KiB of tables, not the cutoff's 192, and 262 144 of
            // them are one launch instead of two. A source above the hint is the generic kernel's: correct, slower)
            size_t lim = c->cdictAttachMax;
            if (sizeHint && sizeHint < lim) { lim = 1024; while (lim < sizeHint) lim <<= 1; }
            if (lim < c->cdictAttachMax) a.slotSrcMax = (uint32_t)lim; else lim = c->cdictAttachMax;
            int w = 10; while (((size_t)1 << w) < lim) w++;
            const int h = c->cdictHlog > w + 1 ? w + 1 : c->cdictHlog, cl = c->cdictClog > w ? w : c->cdictClog;
            stride = (4u << h) + (c->cdictStrat == 2 ? (4u << cl) : 0u);
            if (stride < (4u << 10)) stride = 4u << 10;
            a.arenaLit = (8u * ((uint32_t)lim / 3 + 16) + 15) & ~15u;
            a.arenaStride = (a.arenaLit + (uint32_t)lim + 256 + 15) & ~15u;
        }
        a.tableStride = stride;
        // double-fast without a dictionary: the flat match kernel (one lane per frame, the whole chunk in flight, tables zeroed by a
        // memset) takes every double-fast frame; what it declines goes to the lane-serial kernel through a list. Fast strategy
        // and dictionary batches use the lane-serial kernel for the whole chunk.
        // (r03: dictionary batches whose dictionary row is double-fast take the flat kernel too -- ze_dfast_dict_flat; its waves zero the tables)
        const bool flatDict = c->hasCDict && c->cdictStrat == 2;
        const bool flat = (anyDfast && !c->hasCDict) || flatDict;
        // (round 5) every row double-fast, no dictionary: the flat kernel takes every one-block source of 64 bytes and more and writes sequences only, and what
        // it declines -- sources below 64 bytes, parameter errors -- needs a literal area of its own size at most: the slot is the sequence area + 512 bytes
        // instead of + 128 KiB (21.4 GiB of arena per 65 536 sources instead of 30; it is what lets 262 144 sources be one launch, 88 + 96 GiB)
        bool allDfast = anyDfast && !c->hasCDict; for (int t = 0; t < 4; t++) allDfast = allDfast && a.rows.r[t][6] == 2;
        if (flat && allDfast) a.arenaStride = (uint32_t)(ZE_ARENA_LIT + 512);
        // frames per launch of the flat match kernel: the search is a latency chain per frame, so its rate grows with the frames in flight -- 16 384: 204 ms,
        // 32 768: 270, 65 536: 417, 131 072: 760 (r04za: 9 % less per frame than two launches of 65 536, a second wave per SIMD) -- and what it costs
        // is memory, 384 KiB of tables + 196 KiB of arena per frame. Batches above 65 536 take 131 072 per launch where the device has that free.
        // (the decision is taken once per context -- hipMemGetInfo is not a launch-path call -- and a reservation that fails at 131 072
        // after all, because another context took the memory in between, falls back to 65 536 below instead of failing the call)
        size_t flatMax = c->knob.echunkMax;
        if (!flatMax) {
            flatMax = 65536;
            if (flat && !c->hasCDict && n > 65536) {
                if (!c->flatMaxCached) {
                    size_t freeB = 0, totalB = 0;
                    c->flatMaxCached = 65536;
                    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                        const size_t need = (size_t)131072 * ((size_t)a.tableStride + (size_t)a.arenaStride) / 8 * 9 + ((size_t)4 << 30);
                        if (freeB + c->encFlatTables.cap + c->encArena.cap >= need) c->flatMaxCached = 131072;
                    }
                }
                flatMax = c->flatMaxCached;
            }
        }
        // (the fast strategy -- levels 1, 2, negative -- runs in the lane-serial match kernel, whose time is ONE source's chain of dependent round trips: every source in flight at once
        // is one chain's time, two chunks of 32 768 are two. ZHIP_FAST_WIDE: 65 536 per chunk, sixteen sources per wave -- the kernel's 111 VGPRs hold 4 096 waves)
        const bool fastWide = ZHIP_FAST_WIDE && !flat && !c->hasCDict && n > 32768;
        size_t chunkMax = c->hasCDict ? 262144 : flat ? flatMax : fastWide ? 65536 : 32768;
        if (flat) { const size_t byMem = ((size_t)(flatMax > 65536 ? 96 : 32) << 30) / a.tableStride; if (chunkMax > byMem) chunkMax = byMem; }
        if (c->knob.echunk && c->knob.echunk < chunkMax) chunkMax = c->knob.echunk;
        // Sources of several blocks (the caller's size hint says so) in a double-fast batch without dictionary: the flat kernel searches them
        // too, a lane per frame over all its blocks (ZeMbBlock, zhip_format.hpp); the generic kernel then only does their entropy coding and
        // runs once per chunk. Per frame: block records and room for its sequences (a sequence covers four bytes or more).
        const bool mbc = flat && !flatDict && mbcWanted;
        const size_t mbMaxBlocks = mbc ? 2 * ((sizeHint + ZF_BLOCK_MAX - 1) / ZF_BLOCK_MAX) + 2 : 0;
        const size_t mbSeqCap = mbc ? sizeHint / 4 + mbMaxBlocks + 64 : 0;          // (double-fast matches are four bytes or more: zstd.c:31167 / :31150)
        if (mbc) { const size_t byMem = ((size_t)32 << 30) / (mbSeqCap * 8); if (chunkMax > byMem) chunkMax = byMem ? byMem : 1; }
        const size_t chunk = n < chunkMax ? n : chunkMax;
        const size_t cap = chunk;                                              // items the arenas are sized for
        const size_t laneCap = c->hasCDict ? 262144 : fastWide ? 65536 : 32768;          // lanes of the lane-serial match kernel in flight (each owns tableStride bytes of tables)
        const size_t e1Lanes = c->hasCDict ? ZE_E1_LANES_DICT : fastWide ? 16 : ZE_E1_LANES;
        a.e1Lanes = (uint32_t)e1Lanes;
        size_t g1max = (size_t)c->numCU * (size_t)c->e1PerCU; if (g1max * e1Lanes > laneCap) g1max = laneCap / e1Lanes;
        if (flat && g1max > 256) g1max = 256;                                      // only the frames the flat kernel declines
        const size_t w1 = (chunk + e1Lanes - 1) / e1Lanes;
        const uint32_t g1 = (uint32_t)(w1 < g1max ? w1 : g1max);
        size_t g2max = (size_t)c->numCU * (size_t)c->e2PerCU;
        const uint32_t g2 = (uint32_t)(chunk < g2max ? chunk : g2max);
        const size_t w1cap = (cap + e1Lanes - 1) / e1Lanes, g1cap = w1cap < g1max ? w1cap : g1max, g2cap = cap < g2max ? cap : g2max;
        // waves for what the generic kernel takes: inputs above one block, and -- with a dictionary -- inputs above the attach cutoff
        // (with a dictionary: inputs above the attach cutoff too. Every wave of this kernel carries 544 bytes of scratch per lane: a full-chip
        // grid that finds an empty list still took 2.3 ms of every dictionary batch -- r02zi kernel trace; half a wave per CU is 0.4)
        // (r03: when the caller says its sources exceed one block -- the host API knows, a device-API caller can tell with zhip_ctx_set_size_hint --
        // the list is the whole batch and gets the whole chip: 2 048 x 1 MiB took 10.6 s on 64 waves, profiles/r03_multiblock_rate.txt)
        const size_t gBigMax = sizeHint > ZF_BLOCK_MAX ? (size_t)c->numCU * (size_t)c->encBlocksPerCU : c->hasCDict ? (size_t)c->numCU / 2 : 64;
        const uint32_t gBig = (uint32_t)(n < gBigMax ? n : gBigMax);
        const size_t tabPer = g1cap * e1Lanes * a.tableStride, wsPer = g2cap * ZE_E2_STRIDE + ZHIP_ENC_STRIDE, bigListPer = n * sizeof(uint32_t) + 16,
                     e1ListPer = cap * sizeof(uint32_t) + 16, bigWsPer = (size_t)gBig * ZHIP_ENC_STRIDE;
        if (c->encMeta.reserve(cap * sizeof(ZeMeta)) || c->encArena.reserve(cap * (size_t)a.arenaStride) ||
            c->encTables.reserve(tabPer) || c->encWorkspace.reserve(wsPer) ||
            c->encBigList.reserve(bigListPer) || c->encE1List.reserve(e1ListPer) ||
            (flat && c->encFlatTables.reserve(cap * (size_t)a.tableStride))) {
            if (g_reserveRc == ZHIP_ERR_NO_MEMORY && !c->knob.echunkMax && c->flatMaxCached > 65536) {      // no room for 131 072 per launch after all: 65 536
                c->flatMaxCached = 65536; c->encArena.release(); c->encFlatTables.release();
                return zhip_compress_batch_device(c, d_src, d_srcSegs, n, d_dst, d_dstSegs, d_outSizes, d_status, streamv);
            }
            return g_reserveRc;
        }
        if (mbc) {
            if (c->encMbBlocks.reserve(chunk * mbMaxBlocks * sizeof(ZeMbBlock)) || c->encMbCount.reserve(chunk * sizeof(uint32_t) + 16) ||
                c->encMbSeqs.reserve(chunk * mbSeqCap * 8)) return g_reserveRc;
            a.mbBlocks = (ZeMbBlock*)c->encMbBlocks.p; a.mbCount = (uint32_t*)c->encMbCount.p; a.mbSeqs = (uint64_t*)c->encMbSeqs.p;
            a.mbMaxBlocks = (uint32_t)mbMaxBlocks; a.mbSeqCap = (uint32_t)mbSeqCap; a.mbLanes = c->knob.mbcLanes; a.mbProbes = chunk * ((sizeHint + ZF_BLOCK_MAX - 1) / ZF_BLOCK_MAX) <= c->knob.flat4Max ? 4u : 2u;        // (the traffic in flight is sources x blocks each: 16 384 x 256 KiB 7.9 -> 8.9 GB/s with four probes, 32 768 x 256 KiB 12.0 -> 11.8, r04zj)
            if (c->encBigWs.reserve((size_t)gBig * ZHIP_ENC_STRIDE)) return g_reserveRc;
        }
        a.workspace = (uint8_t*)c->encWorkspace.p;
        a.meta = (ZeMeta*)c->encMeta.p; a.arena = (uint8_t*)c->encArena.p; a.laneTables = (uint8_t*)c->encTables.p;
        uint8_t* flatTables = flat ? (uint8_t*)c->encFlatTables.p : nullptr;
        a.flatTables = flatTables; a.e1List = (uint32_t*)c->encE1List.p; a.e1Count = (uint32_t*)(cbase + 32);
        a.idle = cbase + 256;                                                       // (the context's own counter block: ADVICE r04 -- the caller's arena base need not be readable)
        a.useE1List = flat ? 1u : 0u;
        a.bigList = (uint32_t*)c->encBigList.p; a.bigCount = (uint32_t*)(cbase + 24);
        // Picking the tables' placement (round 5). The flat match kernel's time for one and the same launch differs by ~14 % with WHERE the driver put
        // this 25 GiB allocation (397-415 ms or 455-488 per 65 536 frames; sticky for the life of the allocation, not controllable
        // through the allocator) -- and two allocations held at the same time are different memory. So the first large launch of a context times the
        // real kernel on its tables, reserves a second set beside them, times that, and keeps the faster (tests/tools/e1f_pick_best.py, r05g: in every
        // trial at least one of three candidates was the fast kind, and the best stayed the best). Costs two extra launches of the kernel and a transient
        // second table allocation, once per context; skipped where the second set does not fit. ZHIP_E1F_PICK=0 turns it off.
        // The dictionary search's tables are not zeroed per launch: a cell carries its launch's number above the index and reads as empty under any other number.
        // The index space is 2 + the dictionary's content + a source (+ slack): what is left of 32 bits counts launches, at least 6 bits or the kernel zeroes as before.
        // (the dictionary-less flat search of one-block sources does the same with six fixed bits: its cells are position 18 | tag 8 | launch number 6, ze_dfast_flat_np. `zeroed`
        // comes back false when the tables carry no launch numbers -- the several-block search, an index space too wide -- and the caller zeroes them as before.)
        auto nextEpoch = [&](ZhipEncodeArgs& args, uint8_t* tables, uint64_t gen, size_t bytes, bool* zeroed) -> int {
            args.tabEpoch = 0; args.tabEpochShift = 31; *zeroed = false;
            if (!ZHIP_TABLE_EPOCHS || !flat || mbc) return 0;
            uint32_t es = 26;
            if (flatDict) {
                const uint64_t span = 2ull + c->cdictContentSize + (a.slotSrcMax ? a.slotSrcMax : c->cdictAttachMax) + 64;
                es = 1; while ((1ull << es) < span) es++;
            }
            if (es > 26) return 0;
            *zeroed = true;
            const uint32_t maxE = (1u << (32 - es)) - 1;
            const uint64_t key = (flatDict ? c->cdictKey : 0x9E3779B97F4A7C15ull) ^ ((uint64_t)a.tableStride << 40);
            if (tables != c->encEpochPtr || gen != c->encEpochGen || es != c->encEpochShift || key != c->encEpochKey || c->encEpoch >= maxE || bytes > c->encEpochBytes) {
                HIP_TRY(hipMemsetAsync(tables, 0, bytes, stream));
                c->encEpoch = 0; c->encEpochPtr = tables; c->encEpochGen = gen; c->encEpochShift = es; c->encEpochKey = key; c->encEpochBytes = bytes;
            }
            args.tabEpoch = ++c->encEpoch; args.tabEpochShift = es;
            return 0;
        };
        if (flat && !mbc && c->knob.e1fPick && chunk >= ZHIP_PICK_MIN && (!c->e1fPicked || c->e1fPickedPtr != c->encFlatTables.p)) {      // (again when a larger batch made the context reallocate its tables)
            c->e1fPicked = true; c->e1fPickKept = 0; c->e1fPickMs[1] = c->e1fPickMs[2] = 0;
            const size_t cnt0 = chunk, bytes = cnt0 * (size_t)a.tableStride;
            DevBuf cand;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            struct PickGuard { DevBuf& b; hipEvent_t& x; hipEvent_t& y; ~PickGuard() { b.release(); if (x) (void)hipEventDestroy(x); if (y) (void)hipEventDestroy(y); } } pickGuard{cand, e0, e1};
            if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                ZhipEncodeArgs pa = a; pa.first = 0; pa.count = (uint32_t)cnt0;
                // (round 6, last session: a probe searches the first 8 KiB of every source only. What makes an allocation the slow kind shows in any stretch of the launch -- eight
                // candidates in each of six processes, whole launch / first 8 KiB: 405-412 ms / 30.1-30.5 or 468-485 / 35.7-36.4, never out of order (profiles/r06zzr_pick_study.txt) --
                // so a candidate costs ~35 ms instead of ~450, the first large launch of a context 0.1-0.2 s instead of 1.3, and three slow candidates in a row (one time in eleven
                // at this box's 4 in 9) get three more tries by the rule below instead of a slow context)
                pa.probeCap = flatDict ? 0u : 8192u;
                auto timeOn = [&](uint8_t* t, float* ms) -> int {
                    pa.flatTables = t;
                    HIP_TRY(hipMemsetAsync(cbase + 8, 0, 8, stream)); HIP_TRY(hipMemsetAsync(cbase + 24, 0, 12, stream));
                    {   bool z = false;                                                    // (launch numbers in the cells: a candidate allocation is zeroed once, by nextEpoch)
                        if (int rc = nextEpoch(pa, t, t == (uint8_t*)cand.p ? cand.gen : c->encFlatTables.gen, bytes, &z)) return rc;
                        if (!z && !flatDict) { HIP_TRY(hipMemsetAsync(t, 0, bytes, stream)); c->encEpochPtr = nullptr; } }
                    HIP_TRY(hipEventRecord(e0, stream));
                    launch_flat(!flatDict && c->knob.flat3 && cnt0 <= c->knob.flat3Max ? 3 : 2, cnt0, stream, pa);
                    HIP_TRY(hipEventRecord(e1, stream));
                    HIP_TRY(hipEventSynchronize(e1));
                    HIP_TRY(hipEventElapsedTime(ms, e0, e1));
                    return 0;
                };
                // three candidates: the context's own tables, then two further sets, each reserved beside the best so far (r05g / r05h: 397-415 or 455-488 ms per
                // 65 536 frames; 746-764, 805, 855, 928 per 131 072); the fastest is kept
#if ZHIP_PICK_STUDY
                // DIAGNOSTIC build: eight candidate allocations, each timed whole and over the sources' first 8 KiB; the probe's waves leave their durations (wall clock, 10 ns units), printed
                // as the mean of each sixteenth of the allocation -- is a slow allocation slow everywhere?
                {   const uint32_t cap0 = pa.probeCap;
                    const size_t nw = (cnt0 + 63) / 64;
                    unsigned long long* dClock = nullptr; HIP_TRY(hipMalloc((void**)&dClock, nw * 8));
                    std::vector<unsigned long long> hClock(nw);
                    for (int k = 0; k < (ZHIP_PICK_STUDY > 1 ? ZHIP_PICK_STUDY : 8); k++) {            // (-DZHIP_PICK_STUDY=n, n > 1: n candidates, whole launches for the first eight only)
                        uint8_t* t = flatTables;
                        if (k) { if (cand.reserve(bytes)) { (void)hipGetLastError(); break; } t = (uint8_t*)cand.p; }
                        float whole = 0, probe = 0;
                        pa.probeCap = 0; if (k < 8) if (int rc = timeOn(t, &whole)) return rc;
                        HIP_TRY(hipMemsetAsync(dClock, 0, nw * 8, stream));
                        pa.probeCap = 8192; pa.waveClock = dClock; if (int rc = timeOn(t, &probe)) return rc;
                        pa.waveClock = nullptr; pa.probeCap = cap0;
                        HIP_TRY(hipMemcpy(hClock.data(), dClock, nw * 8, hipMemcpyDeviceToHost));
                        char line[512]; int o = 0; unsigned long long mn = ~0ull, mx = 0;
                        for (int q = 0; q < 16; q++) { double sum = 0; size_t lo = nw * q / 16, hi = nw * (q + 1) / 16; for (size_t w = lo; w < hi; w++) { sum += (double)hClock[w]; if (hClock[w] < mn) mn = hClock[w]; if (hClock[w] > mx) mx = hClock[w]; }
                                                       o += snprintf(line + o, sizeof line - o, " %.2f", sum / (double)(hi - lo ? hi - lo : 1) / 1e5); }
                        fprintf(stderr, "[zhip-pick-study] candidate %d %p whole %.1f ms, first 8 KiB %.1f; waves' ms by sixteenth of the allocation:%s (min %.2f max %.2f)\n", k, (void*)t, whole, probe, line, mn / 1e5, mx / 1e5);
                        if (k) cand.release();
                    }
                    (void)hipFree(dClock);
                }
#endif
                float best = 0;
                if (int rc = timeOn(flatTables, &best)) return rc;
                c->e1fPickMs[0] = best;
                // (round 6: where a probe is cheap -- dictionary batches: ~40 ms and 12 GiB a candidate -- and the first three came out ALIKE, which says they are one kind but not
                // which, up to three more are tried until one is clearly faster: a dictionary batch's candidates are the fast kind one time in three, 35.4 ms against 41-42
                // (r06u: [35.6, 42.1, 42.1], [42.3, 35.3, 42.1]; r06z: [41.1, 40.9, 41.2] -- all slow, 23.2 GB/s instead of 26.5), so three of them are all slow three runs in ten)
                // (round 6, last session: EIGHT candidates, always. With probes at ~35 ms there is no reason to stop at the first fast one, and there is a third kind: the wave-clock study
                // (profiles/r06zzv_wave_clocks.txt) found allocations at 388-392 ms per 65 536 sources beside the 406-417 and 467-473 ones -- one candidate in eight, the eighth in each of
                // three processes; an allocation's kind is the same over all of its sixteenths, so there is nothing to pick INSIDE one)
                // (a released candidate's memory does not come back at once: the EIGHTH 28 GiB reservation of a 65 536-source context, with ~70 GiB of the process's own in use, waited
                // 4.7 s for the driver to reclaim what the earlier candidates had left -- profiles/r06zzz6_pick_candidates_cost.txt -- where every other one took 0.3 ms. So the candidates'
                // reservations together stay inside what was free when the pick began, less one more set and 8 GiB: seven candidates there, two at 131 072 sources per launch)
                float worst = best;
                size_t free0 = 0, total0 = 0;
                if (hipMemGetInfo(&free0, &total0) != hipSuccess) { (void)hipGetLastError(); free0 = 0; }
                const size_t want = bytes + (bytes >> 3) + 4096;
                for (int k = 1; k < ZHIP_PICK_CANDIDATES; k++) {
                    if ((size_t)k * want + want + ((size_t)8 << 30) > free0) break;
                    const auto tc0 = std::chrono::steady_clock::now();
                    if (cand.reserve(bytes)) { (void)hipGetLastError(); break; }                 // no room for another set: keep what we have
                    const auto tc1 = std::chrono::steady_clock::now();
                    float ms = 0;
                    if (int rc = timeOn((uint8_t*)cand.p, &ms)) return rc;
                    if (c->knob.prof) fprintf(stderr, "[zhip-prof] pick candidate %d: reserve %.1f ms, zero + probe %.1f ms (the probe launch %.1f)\n", k,
                                              std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count(), ms);
                    if (k < 3) c->e1fPickMs[k] = ms;
                    if (ms > worst) worst = ms;
                    if (ms < 0.97f * best) { std::swap(c->encFlatTables.p, cand.p); std::swap(c->encFlatTables.cap, cand.cap); std::swap(c->encFlatTables.gen, cand.gen); best = ms; c->e1fPickKept = k; }
                    cand.release();
                    // (no early stop among the first three: at 131 072 frames per launch the kinds are not two but a spread -- 750, 805, 855, 928 ms, r05n / r05u -- so all three are timed)
                }
                flatTables = (uint8_t*)c->encFlatTables.p; a.flatTables = flatTables;
            } else (void)hipGetLastError();
            c->e1fPickedPtr = c->encFlatTables.p;
        }
        HIP_TRY(hipMemsetAsync(cbase + 24, 0, 8, stream));
        if (c->knob.prof) {                                                         // tuning aid: per-phase cycle totals of the entropy kernel
            if (!c->profEncode) HIP_TRY(hipMalloc((void**)&c->profEncode, 16 * 8));
            HIP_TRY(hipMemsetAsync(c->profEncode, 0, 16 * 8, stream));
            a.prof = c->profEncode;
        }
        for (size_t first = 0; first < n; first += chunk) {
            const size_t cnt = n - first < chunk ? n - first : chunk;
            a.first = (uint32_t)first; a.count = (uint32_t)cnt;
            HIP_TRY(hipMemsetAsync(cbase + 8, 0, 8, stream));
            HIP_TRY(hipMemsetAsync(cbase + 32, 0, 4, stream));
            const bool tm = c->timing;
            hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (tm) for (int i = 0; i < 6; i++) HIP_TRY(hipEventCreate(&ev[i]));
            if (flat) {
                // every launch its own number in the tables' cells (the tables serve other sources now), or -- where the cells carry none -- zeroed tables
                bool z = false;
                if (int rc = nextEpoch(a, flatTables, c->encFlatTables.gen, cap * (size_t)a.tableStride, &z)) return rc;
                if (!z && !flatDict) { HIP_TRY(hipMemsetAsync(flatTables, 0, cnt * (size_t)a.tableStride, stream)); c->encEpochPtr = nullptr; }      // (cells without numbers in the tables now: the next numbered launch starts from zeroed tables)
                if (tm) HIP_TRY(hipEventRecord(ev[0], stream));
                // a few frames: nothing hides the search's round trips, so each frame gets a CU and its source goes to LDS (ze_match_lds_body)
                // (the LDS area follows the batch's largest source where the caller told us -- the host-buffer API does: more frames per CU)
                const size_t hint = c->srcMaxHint ? c->srcMaxHint : (size_t)ZF_BLOCK_MAX;
                const int shape = hint <= 4096 ? 0 : hint <= 16384 ? 1 : hint <= 65536 ? 2 : 3;
                static const unsigned wavesPerCU[4] = { 32, 9, 2, 1 };          // by LDS (160 KiB per CU) and the 32-wave limit
                const size_t rounds = shape == 3 ? ZHIP_E1LDS_PER_CU : c->knob.e1LdsRounds;
                const size_t ldsMax = c->knob.e1LdsMax >= 0 ? (size_t)c->knob.e1LdsMax : (size_t)c->numCU * wavesPerCU[shape] * rounds;
                if (mbc) hipLaunchKernelGGL(zhip_encode_split_kernel, dim3((uint32_t)(cnt < (size_t)c->numCU * 8 ? cnt : (size_t)c->numCU * 8)), dim3(64), 0, stream, a);
                if (cnt <= ldsMax && !flatDict && !mbc) {
                    const dim3 g((uint32_t)cnt), b(64);
                    if (shape == 0) hipLaunchKernelGGL((zhip_encode_match_lds_kernel<4096, ZHIP_E1LDS_PROBES>), g, b, 0, stream, a);
                    else if (shape == 1) hipLaunchKernelGGL((zhip_encode_match_lds_kernel<16384, ZHIP_E1LDS_PROBES>), g, b, 0, stream, a);
                    else if (shape == 2) hipLaunchKernelGGL((zhip_encode_match_lds_kernel<65536, ZHIP_E1LDS_PROBES>), g, b, 0, stream, a);
                    else hipLaunchKernelGGL((zhip_encode_match_lds_kernel<ZF_BLOCK_MAX, ZHIP_E1LDS_PROBES>), g, b, 0, stream, a);
                }
                else launch_flat(!flatDict && !mbc && cnt <= c->knob.flat4Max ? 4 : !flatDict && !mbc && c->knob.flat3 && cnt <= c->knob.flat3Max ? 3 : 2, cnt, stream, a);
                if (mbc) hipLaunchKernelGGL(zhip_encode_match_flat_mb_kernel, dim3((uint32_t)((cnt + a.mbLanes - 1) / a.mbLanes)), dim3(64), 0, stream, a);
                if (tm) HIP_TRY(hipEventRecord(ev[1], stream));
            }
            if (tm) HIP_TRY(hipEventRecord(ev[2], stream));
            hipLaunchKernelGGL(zhip_encode_match_kernel, dim3(g1), dim3(64), 0, stream, a);
            if (tm) { HIP_TRY(hipEventRecord(ev[3], stream)); HIP_TRY(hipEventRecord(ev[4], stream)); }
            a.xxLater = ZHIP_TRAILER_LATER && a.checksumFlag ? 1u : 0u;
            hipLaunchKernelGGL(zhip_encode_entropy_kernel, dim3(g2), dim3(64), 0, stream, a);
            if (a.xxLater) { const size_t w = (cnt + 63) / 64, gm = (size_t)c->numCU * 8; hipLaunchKernelGGL(zhip_encode_trailer_kernel, dim3((uint32_t)(w < gm ? w : gm)), dim3(64), 0, stream, a); }      // EX (timed with E2)
            a.xxLater = 0;
            if (tm) HIP_TRY(hipEventRecord(ev[5], stream));
            if (mbc) {      // this chunk's sources of several blocks: the generic kernel over the list the flat kernel just made (it reads the chunk's arenas)
                ZhipEncodeArgs b = a;
                b.workspace = (uint8_t*)c->encBigWs.p; b.counter = (uint32_t*)(cbase + 28);
                b.frameList = a.bigList; b.listCount = a.bigCount;
                hipLaunchKernelGGL(zhip_encode_frames_kernel, dim3(gBig), dim3(64), 0, stream, b);
                HIP_TRY(hipMemsetAsync(cbase + 24, 0, 8, stream));          // list length and the kernel's work counter: fresh for the next chunk
            }
            HIP_TRY(hipGetLastError());
            if (tm) {
                if (flat) c->timer[8].pending.emplace_back(ev[0], ev[1]);
                else { (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]); }
                c->timer[5].pending.emplace_back(ev[2], ev[3]);
                c->timer[6].pending.emplace_back(ev[4], ev[5]);
            }
        }
        if (a.prof) {
            HIP_TRY(hipStreamSynchronize(stream));
            unsigned long long h[16];
            HIP_TRY(hipMemcpy(h, a.prof, sizeof h, hipMemcpyDeviceToHost));
            static const char* nm[12] = {"gather literals", "literal stats+decide", "huffman build+table", "huffman encode", "sequence stats", "sequence tables", "sequence stream", "frame assembly",
                                          "  stream: constants", "  stream: state chains", "  stream: pack + OR", "  stream: flush"};      // (the last four: -DZE_PROF_STREAM builds only)
            unsigned long long tot = 0; for (int q = 0; q <= ZEP_REST; q++) tot += h[q];
            fprintf(stderr, "[zhip-prof] E2: %.0f wave-cycles per frame\n", (double)tot / (double)n);
            for (int q = 0; q < ZEP_N; q++) fprintf(stderr, "[zhip-prof]    %-22s %6.2f%% %10.0f cyc/frame\n", nm[q], 100.0 * h[q] / (tot ? tot : 1), (double)h[q] / (double)n);
            a.prof = nullptr;
        }
        if (!mbc) {   // inputs above 128 KiB (multi-block frames): the generic one-wave-per-frame kernel over the list E1 made (usually empty)
            if (c->encBigWs.reserve(bigWsPer)) return g_reserveRc;
            ZhipEncodeArgs b = a;
            b.workspace = (uint8_t*)c->encBigWs.p; b.counter = (uint32_t*)(cbase + 28);
            b.frameList = a.bigList; b.listCount = a.bigCount;
            hipLaunchKernelGGL(zhip_encode_frames_kernel, dim3(gBig), dim3(64), 0, stream, b);
            HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    if (c->encWorkspace.reserve((size_t)grid * ZHIP_ENC_STRIDE)) return g_reserveRc;
    a.workspace = (uint8_t*)c->encWorkspace.p;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    hipLaunchKernelGGL(zhip_encode_frames_kernel, dim3(grid), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    if (c->timing) { HIP_TRY(hipEventRecord(e1, stream)); c->timer[1].pending.emplace_back(e0, e1); }
    if (c->timer[1].pending.size() > 4096) { HIP_TRY(hipStreamSynchronize(stream)); drain_timer(c->timer[1]); }
    return 0;
}

extern "C" int zhip_ctx_sync(zhip_ctx* c, void* streamv, const int32_t* d_status, size_t n, zhip_error* err)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)streamv));
    if (err) memset(err, 0, sizeof *err);
    if (!d_status || !n) return 0;
    std::vector<int32_t> st(n);
    HIP_TRY(hipMemcpy(st.data(), d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) if (st[i]) {
        if (err) { err->kind = ZHIP_ERR_ZSTD; err->zstdErr = st[i]; err->index = i; }
        return ZHIP_ERR_ZSTD;
    }
    (void)c;
    return 0;
}

// ------------------------------------------------------------------------------------------ host-buffer batch API
// zhip_compress_batch / zhip_decompress_batch: the reference's multi_*_to_buffer hands over host buffers and gets host buffers back
// (c-ext/compressor.c:1340-1503, c-ext/decompressor.c:1459-1710), so this path is bounded by PCIe (Gen5 x16, ~55 GB/s each way) --
// provided nothing on the host is slower than the link. Round 1 was: one thread packs every item into a pinned buffer, ONE H2D copy,
// kernels, ONE D2H copy into pageable memory (10.5 / 2.6 GB/s). Now the batch is cut into chunks that flow through three streams:
//
//     host threads pack chunk k+1 into pinned staging  |  H2D(k+1)  |  kernels(k)  |  D2H(k-1) straight into the result payload
//
// * staging: two pinned slots, filled by ZHIP_PACK_THREADS host threads (a pageable->pinned memcpy is ~8 GB/s per thread);
// * results: one zhip_outbuf per chunk (the reference returns one buffer per worker as well, the grouping is not contractual), its
//   payload PINNED so that the D2H copy runs at link speed and needs no second host copy. Pinning costs ~0.3 ms per MiB, far more than
//   the copy itself, so payload blocks come from a process-wide pool and return to it when the caller frees them
//   (zhip_free_payload; the CPython extension's BufferWithSegments does that in its deallocator);
// * compress: the compressBound-sized slots are compacted ON THE DEVICE (scan of the frame sizes + one wave per frame) so that only
//   the frames cross the link, into a payload allocated once the chunk's total is known (the host learns it one chunk behind).
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#ifndef ZHIP_PACK_THREADS
#define ZHIP_PACK_THREADS 8
#endif
#define ZHIP_HOST_CHUNK_ITEMS 32768u                  // most items of one pipeline chunk (the scan kernel's bound)
#define ZHIP_PIN_MIN ((size_t)1 << 20)                // payloads below 1 MiB are plain malloc() (one-shot calls, small batches)
// idle pinned bytes kept for reuse; beyond that blocks are unpinned on free. 24 GiB covers a full-size batch's payloads (pinning costs more than
// the copy it speeds up); ZHIP_PIN_POOL_KEEP_MB (read once) lowers or raises it, 0 = keep nothing (no way to trim it before)
static size_t pin_pool_keep()
{
    static const size_t keep = [] { const char* e = getenv("ZHIP_PIN_POOL_KEEP_MB"); return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)24 << 30; }();
    return keep;
}
#define ZHIP_PIN_POOL_KEEP pin_pool_keep()

static std::atomic<bool> g_pinPortable{false};      // set when the host-buffer calls have more than one device slot to fan out over (DevPool::parse)
namespace {
struct PinBlock { size_t cap; bool busy; };
struct PinPool {
    std::mutex mu;
    std::unordered_map<void*, PinBlock> blocks;
    size_t idle = 0;
    void* take(size_t n)
    {
        if (n < ZHIP_PIN_MIN) return malloc(n ? n : 1);
        {
            std::lock_guard<std::mutex> g(mu);
            void* best = nullptr; size_t bestCap = ~(size_t)0;
            for (auto& kv : blocks) if (!kv.second.busy && kv.second.cap >= n && kv.second.cap < bestCap) { best = kv.first; bestCap = kv.second.cap; }
            if (best && bestCap <= n + (n >> 1) + ((size_t)64 << 20)) { blocks[best].busy = true; idle -= bestCap; return best; }
        }
        void* p = nullptr;
        const size_t cap = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        // (portable -- pinned for every device's copy engines -- only where the in-call fan-out can hand a block to another device: fan_out sets the flag)
        if (hipHostMalloc(&p, cap, g_pinPortable.load() ? hipHostMallocPortable : hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return malloc(n); }     // pageable still works, slower
        std::lock_guard<std::mutex> g(mu);
        blocks[p] = PinBlock{cap, true};
        return p;
    }
    void give(void* p)
    {
        if (!p) return;
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = blocks.find(p);
            if (it == blocks.end()) { free(p); return; }
            it->second.busy = false; idle += it->second.cap;
            while (idle > ZHIP_PIN_POOL_KEEP) {                       // unpin the largest idle blocks first
                void* big = nullptr; size_t bigCap = 0;
                for (auto& kv : blocks) if (!kv.second.busy && kv.second.cap > bigCap) { big = kv.first; bigCap = kv.second.cap; }
                if (!big) break;
                idle -= bigCap; blocks.erase(big); drop.push_back(big);
            }
        }
        for (void* q : drop) (void)hipHostFree(q);
    }
};
PinPool& pin_pool() { static PinPool* pool = new PinPool(); return *pool; }      // leaked on purpose: payloads may outlive static destruction
}

extern "C" void zhip_free_payload(void* p) { pin_pool().give(p); }

// sizes -> exclusive offsets (bytes) of the frames of one chunk, total to *total. One workgroup; n <= ZHIP_HOST_CHUNK_ITEMS.
__global__ __launch_bounds__(1024) void zhip_scan_sizes_kernel(const uint64_t* sizes, const int32_t* status, uint32_t n, uint64_t* offs, uint64_t* total)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023) / 1024, lo = t * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += status[i] ? 0 : sizes[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) { offs[i] = run; run += status[i] ? 0 : sizes[i]; }
    if (t == 1023) *total = part[1023];
}
// one wave per frame: the valid prefix of its compressBound-sized slot -> its place in the dense payload
__global__ __launch_bounds__(64) void zhip_compact_kernel(const uint8_t* slots, const zhip_segment* dstSegs, const uint64_t* sizes, const int32_t* status,
                                                           const uint64_t* offs, uint32_t n, uint8_t* dense)
{
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (status[i]) continue;
        const uint8_t* s = slots + dstSegs[i].offset; uint8_t* d = dense + offs[i];
        const uint32_t size = (uint32_t)sizes[i], whole = size & ~15u;
        for (uint32_t j = threadIdx.x * 16; j < whole; j += 1024) { const zh_v16 v = zh_ld128(s + j); zh_st64(d + j, v.lo); zh_st64(d + j + 8, v.hi); }
        if (threadIdx.x < size - whole) d[whole + threadIdx.x] = s[whole + threadIdx.x];
    }
}

extern "C" void zhip_ctx_set_size_hint(zhip_ctx* c, uint64_t maxItemBytes) { if (c) c->itemHint = (size_t)maxItemBytes; }

extern "C" int zhip_compact_device(const void* d_slots, const zhip_segment* d_slotSegs, const uint64_t* d_outSizes, const int32_t* d_status,
                                   const uint64_t* d_offsets, size_t n, void* d_dense, void* streamv)
{
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFu) { g_lastError = "too many frames in one launch"; return ZHIP_ERR_UNSUPPORTED; }
    // (the CU count per device is looked up once: hipGetDeviceProperties is slow and this sits on the sharded compress path)
    static thread_local int cuOf[64] = {0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int cus = dev >= 0 && dev < 64 ? cuOf[dev] : 0;
    if (!cus) { HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); if (dev >= 0 && dev < 64) cuOf[dev] = cus; }
    const size_t gmax = (size_t)cus * 16;
    hipLaunchKernelGGL(zhip_compact_kernel, dim3((uint32_t)(n < gmax ? n : gmax)), dim3(64), 0, (hipStream_t)streamv, (const uint8_t*)d_slots, d_slotSegs, d_outSizes,
                       d_status, d_offsets, (uint32_t)n, (uint8_t*)d_dense);
    HIP_TRY(hipGetLastError());
    return 0;
}

// per-context host pipeline state (streams, staging, small pinned metadata)
static int host_pipe_init(zhip_ctx* c)
{
    if (c->hpReady) return 0;
    HIP_TRY(hipStreamCreateWithFlags(&c->hpH2D, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->hpCompute, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->hpD2H, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&c->hpStageFree[i], hipEventDisableTiming));
    c->hpReady = true;
    return 0;
}
static int ensure_stage(zhip_ctx* c, int slot, size_t n)
{
    if (n <= c->hpStageCap[slot]) return 0;
    if (c->hpStage[slot]) (void)hipHostFree(c->hpStage[slot]);
    c->hpStage[slot] = nullptr; c->hpStageCap[slot] = 0;
    const size_t want = n + (n >> 3) + 4096;
    HIP_TRY(hipHostMalloc(&c->hpStage[slot], want, hipHostMallocDefault));      // (write-combined / non-coherent staging: no difference, profiles/r05zb_*)
    c->hpStageCap[slot] = want;
    return 0;
}
static int ensure_pinned(zhip_ctx* c, size_t n)          // small pinned area for per-item sizes / status / chunk totals coming back
{
    if (n <= c->pinnedCap) return 0;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinnedCap = 0;
    size_t want = n + (n >> 2) + 4096;
    HIP_TRY(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    c->pinnedCap = want;
    return 0;
}
// items [lo, hi) -> stage, item i at offset segs[i].offset - segs[lo].offset. Big chunks are split over host threads by bytes.
static void pack_items(uint8_t* stage, const zhip_item* items, const zhip_segment* segs, size_t lo, size_t hi, unsigned packThreads)
{
    const uint64_t base = segs[lo].offset, bytes = hi > lo ? segs[hi - 1].offset + segs[hi - 1].length - base : 0;
    auto run = [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) if (items[i].srcSize) memcpy(stage + (segs[i].offset - base), items[i].src, items[i].srcSize); };
    unsigned nt = bytes >= ((uint64_t)32 << 20) ? (packThreads ? packThreads : ZHIP_PACK_THREADS) : 1;
    if (nt <= 1 || hi - lo < 2 * nt) { run(lo, hi); return; }
    std::vector<std::thread> th;
    size_t a = lo;
    for (unsigned t = 0; t < nt && a < hi; t++) {
        const uint64_t target = base + bytes * (t + 1) / nt;
        size_t b = a;
        while (b < hi && (t + 1 == nt || segs[b].offset + segs[b].length <= target)) b++;
        if (b == a) b = a + 1;
        th.emplace_back(run, a, b);
        a = b;
    }
    for (auto& t : th) t.join();
}
// one lazily created context per calling thread, destroyed when the thread exits; re-created when the thread switched devices
struct TlsCtx { zhip_ctx* c = nullptr; ~TlsCtx() { if (c) zhip_ctx_destroy(c); } };
static thread_local TlsCtx g_tls;
static zhip_ctx* tls_ctx()
{
    int dev = -1;
    if (g_tls.c && hipGetDevice(&dev) == hipSuccess && dev != g_tls.c->device) { zhip_ctx_destroy(g_tls.c); g_tls.c = nullptr; }
    if (!g_tls.c) g_tls.c = zhip_ctx_create();
    return g_tls.c;
}
// the host-buffer API keeps its scratch between calls (allocation is slow), but not the tens of GiB a 65 536-frame batch needs
// what memory_size() reports. The reference's contexts exist from the constructor on (ZSTD_sizeof_CCtx > 0 right away, its tests expect that):
// the calling thread's device context is created here if it is not there yet, with the launch counters every call needs.
static std::atomic<size_t> g_slotBytes[64];         // device bytes the device slots' contexts hold (each slot's thread updates its own after a job: fan_out)
extern "C" size_t zhip_thread_memory_size(void)
{
    zhip_ctx* c = tls_ctx();
    if (c) (void)c->counter.reserve(64 * 8);
    size_t n = c ? c->device_bytes() : 0;
    for (int i = 0; i < 64; i++) n += g_slotBytes[i].load();     // (what the in-call fan-out keeps on the node's devices belongs to the caller's picture too)
    return n;
}
// Round 6: a context KEEPS its working set between calls -- the flat search's tables (12 GiB for a 32 768-source chunk), the arenas, the
// device-side staging -- as long as the whole stays within ZHIP_KEEP_GB (default 64: a BASELINE-sized call of 65 536 x 128 KiB holds ~50 GiB of the GPU's
// 288), the way the reference keeps one ZSTD_CCtx per worker between calls (c-ext/compressor.c:1129-1168). Rounds 1-5 released every buffer above 2 GiB after
// every call: the next call re-reserved ~13 GiB of tables (and re-drew their placement). Above the limit the largest buffers go first.
static size_t tls_keep_bytes()
{
    static const size_t keep = [] { const char* e = getenv("ZHIP_KEEP_GB"); return (e ? (size_t)strtoull(e, nullptr, 10) : (size_t)64) << 30; }();
    return keep;
}
static void tls_trim(zhip_ctx* c)
{
    DevBuf* bufs[] = { &c->pipeMeta, &c->pipeLit, &c->pipeFse, &c->pipeHuf, &c->pipeBases, &c->encArena, &c->encTables, &c->encFlatTables, &c->encWorkspace,
                       &c->encBigWs, &c->scratch, &c->hSrc, &c->hDst, &c->hDense };
    while (c->device_bytes() > tls_keep_bytes()) {
        DevBuf* big = nullptr;
        for (DevBuf* b : bufs) if (b->cap > ((size_t)64 << 20) && (!big || b->cap > big->cap)) big = b;
        if (!big) break;
        big->release();
    }
}
static int set_err(zhip_error* err, int kind, size_t index, int zerr, uint64_t d0 = 0, uint64_t d1 = 0)
{
    if (err) { err->kind = kind; err->index = index; err->zstdErr = zerr; err->detail[0] = d0; err->detail[1] = d1; }
    return kind;
}

extern "C" void zhip_free_outbufs(zhip_outbuf* bufs, size_t n, int freePayload)
{
    if (!bufs) return;
    if (freePayload) for (size_t i = 0; i < n; i++) { zhip_free_payload(bufs[i].data); free(bufs[i].segs); }
    free(bufs);
}

// chunk boundaries [cut[k], cut[k+1]) over n items: a chunk closes when its input or output bytes reach maxBytes or it holds maxItems
// items. segs: [0,n) source, [n,2n) destination.
// firstItems (0: like the others): a smaller first chunk shortens the pipeline's fill -- the time before the first kernel can start
static std::vector<size_t> host_chunks(const zhip_segment* segs, size_t n, uint64_t maxBytes, size_t maxItems, size_t firstItems = 0)
{
    std::vector<size_t> cut(1, 0);
    uint64_t in = 0, out = 0; size_t cnt = 0;
    for (size_t i = 0; i < n; i++) {
        in += segs[i].length; out += segs[n + i].length; cnt++;
        const size_t lim = cut.size() == 1 && firstItems ? firstItems : maxItems;
        if (in >= maxBytes || out >= maxBytes || cnt >= lim) { cut.push_back(i + 1); in = out = 0; cnt = 0; }
    }
    if (cut.back() != n) cut.push_back(n);
    return cut;
}
// items [lo, hi) -> their place in the device source arena, in steps of <= ZHIP_STAGE_BYTES through the two pinned staging slots: the
// packing of step s+1 (host threads) overlaps the H2D copy of step s. Every copy is only ENQUEUED on hpH2D; `done` is recorded behind
// the last one. A slot is reused once the copy that last read it has finished (its event).
#define ZHIP_STAGE_BYTES ((uint64_t)512 << 20)
static int upload_items(zhip_ctx* c, const zhip_item* items, const zhip_segment* segs, size_t lo, size_t hi, hipEvent_t done)
{
    size_t a = lo;
    while (a < hi) {
        size_t b = a; uint64_t bytes = 0;
        while (b < hi && (b == a || bytes + segs[b].length <= ZHIP_STAGE_BYTES)) { bytes += segs[b].length; b++; }
        const int slot = c->hpNextSlot; c->hpNextSlot ^= 1;
        if (hipEventSynchronize(c->hpStageFree[slot]) != hipSuccess) return ZHIP_ERR_HIP;
        if (ensure_stage(c, slot, bytes + 16)) return ZHIP_ERR_HIP;
        pack_items((uint8_t*)c->hpStage[slot], items, segs, a, b, c->knob.packThreads);
        if (bytes && hipMemcpyAsync((uint8_t*)c->hSrc.p + segs[a].offset, c->hpStage[slot], bytes, hipMemcpyHostToDevice, c->hpH2D) != hipSuccess) return ZHIP_ERR_HIP;
        if (hipEventRecord(c->hpStageFree[slot], c->hpH2D) != hipSuccess) return ZHIP_ERR_HIP;
        a = b;
    }
    return hipEventRecord(done, c->hpH2D) == hipSuccess ? 0 : ZHIP_ERR_HIP;
}
static void empty_outbuf(zhip_outbuf* ob) { ob->data = malloc(1); ob->segs = (zhip_segment*)malloc(sizeof(zhip_segment)); ob->dataSize = 0; ob->nSegs = 0; }

// ONE device (the calling thread's current one): the chunked three-stream pipeline
static int decompress_batch_one(const zhip_dparams* params, const zhip_item* items, size_t n, int requireSizes,
                                zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    const bool allowShort = (requireSizes & 2) != 0;   // dstSize is a capacity (one-shot decompress with max_output_size)
    if (err) memset(err, 0, sizeof *err);
    *out = nullptr; *nOut = 0;
    zhip_ctx* c = tls_ctx();
    if (!c) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    // pass 1 (decompress_worker pass 1, decompressor.c:981-1014): every frame needs a known decompressed size
    std::vector<zhip_segment> segs(2 * n);           // [0,n) source, [n,2n) destination
    std::vector<uint32_t> nBlocks(n);                // blocks per frame (0: unknown), from the block headers (host_count_blocks)
    const int format = params ? params->format : ZHIP_FORMAT_ZSTD1;
    uint64_t srcTotal = 0, dstTotal = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t ds = items[i].dstSize;
        if (ds == 0) {
            uint64_t fcs = zhip_frame_content_size_format(items[i].src, items[i].srcSize, format);
            if (fcs == ZHIP_CONTENTSIZE_ERROR || fcs == ZHIP_CONTENTSIZE_UNKNOWN) return set_err(err, ZHIP_ERR_UNKNOWN_SIZE, i, 0);
            ds = fcs;
        }
        // the sizes come from untrusted frame headers: a claimed size that cannot be allocated is "out of memory" (what the reference's
        // malloc of the destination reports, decompressor.c:1085-1090), never a wrapped sum
        if (ds > ((uint64_t)1 << 46) || dstTotal + ds > ((uint64_t)1 << 46)) return set_err(err, ZHIP_ERR_NO_MEMORY, i, 0);
        segs[i].offset = srcTotal; segs[i].length = items[i].srcSize; srcTotal += items[i].srcSize;
        segs[n + i].offset = dstTotal; segs[n + i].length = ds; dstTotal += ds;
        nBlocks[i] = host_count_blocks((const uint8_t*)items[i].src, items[i].srcSize, format);
    }
    int r = zhip_ctx_set_dformat(c, format, params ? params->maxWindowSize : 0);
    if (r) return set_err(err, r, 0, 0);
    r = zhip_ctx_set_ddict(c, params ? params->dict : nullptr, params ? params->dictSize : 0, params ? params->dictType : ZHIP_DICT_AUTO);
    if (r < 0) return set_err(err, ZHIP_ERR_ZSTD, 0, -r);
    if (r) return set_err(err, r, 0, 0);
    if (host_pipe_init(c)) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    // the kernels are several times faster than the link here: chunks of ~1 GiB of output keep all three streams busy
    // (a small FIRST chunk -- ZHIP_HCHUNK_D0 items, default 2 048 for batches above 8 192 -- starts the copy back, which is what the call waits for, a few ms after
    // the call instead of after a whole chunk's packing, upload and kernels)
    const std::vector<size_t> cut = host_chunks(segs.data(), n, (uint64_t)1 << 30, 32768, n > 8192 ? c->knob.hchunkD0 : 0);
    const size_t nChunks = cut.size() - 1;
    if (c->hSrc.reserve(srcTotal + 16) || c->hDst.reserve(dstTotal + 16) || c->hSegs.reserve(2 * n * sizeof(zhip_segment) + 16) ||
        c->hStatus.reserve(n * (sizeof(uint64_t) + sizeof(int32_t)) + 16)) return set_err(err, g_reserveRc, 0, 0);
    if (ensure_pinned(c, n * (sizeof(uint64_t) + sizeof(int32_t)) + 16)) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    zhip_outbuf* ob = (zhip_outbuf*)calloc(nChunks ? nChunks : 1, sizeof(zhip_outbuf));
    if (!ob) return set_err(err, ZHIP_ERR_NO_MEMORY, 0, 0);
    hipEvent_t evUp = nullptr, evK = nullptr;
    auto fail = [&](int kind, size_t index = 0, int zerr = 0, uint64_t d0 = 0, uint64_t d1 = 0) -> int {
        if (kind == ZHIP_ERR_HIP) hip_fail(hipGetLastError(), "host decompress pipeline");
        (void)hipDeviceSynchronize();
        if (evUp) (void)hipEventDestroy(evUp);
        if (evK) (void)hipEventDestroy(evK);
        zhip_free_outbufs(ob, nChunks ? nChunks : 1, 1);
        return set_err(err, kind, index, zerr, d0, d1);
    };
    if (hipEventCreateWithFlags(&evUp, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&evK, hipEventDisableTiming) != hipSuccess) return fail(ZHIP_ERR_HIP);
    if (hipMemcpyAsync(c->hSegs.p, segs.data(), 2 * n * sizeof(zhip_segment), hipMemcpyHostToDevice, c->hpH2D) != hipSuccess ||
        hipStreamSynchronize(c->hpH2D) != hipSuccess) return fail(ZHIP_ERR_HIP);      // segs is pageable: wait before anything reads the table
    uint64_t* dSizes = (uint64_t*)c->hStatus.p;
    int32_t* dStatus = (int32_t*)((uint8_t*)c->hStatus.p + n * sizeof(uint64_t));
    uint64_t* hSizes = (uint64_t*)c->pinned;
    int32_t* hStatus = (int32_t*)((uint8_t*)c->pinned + n * sizeof(uint64_t));
    const zhip_segment* dSegs = (const zhip_segment*)c->hSegs.p;
    for (size_t k = 0; k < nChunks; k++) {
        const size_t lo = cut[k], hi = cut[k + 1], cnt = hi - lo;
        const uint64_t outBytes = segs[n + hi - 1].offset + segs[n + hi - 1].length - segs[n + lo].offset;
        if (upload_items(c, items, segs.data(), lo, hi, evUp)) return fail(ZHIP_ERR_HIP);
        if (hipStreamWaitEvent(c->hpCompute, evUp, 0) != hipSuccess) return fail(ZHIP_ERR_HIP);
        {   uint64_t mx = 0, slotsWanted = 0; size_t several = 0;
            for (size_t i = lo; i < hi; i++) {
                const uint64_t len = segs[n + i].length;
                if (len > mx) mx = len;
                // (a frame's block count is known where its block headers could be walked: the slots it will take; else the estimate from its size)
                slotsWanted += nBlocks[i] > 1 ? (uint64_t)nBlocks[i] + 1 : len <= ZF_BLOCK_MAX ? 1 : 2 * ((len + ZF_BLOCK_MAX - 1) / ZF_BLOCK_MAX) + 2;
                several += nBlocks[i] > 1 && len <= ZF_BLOCK_MAX;
            }
            // frames of one block's size cut into SEVERAL blocks (the block splitter of levels 16+): from one in sixteen on the chunk takes the several-block mode -- a size hint just
            // above a block says so --, below that the few go to the generic kernel as before
            if (several * 16 >= cnt && mx <= ZF_BLOCK_MAX) mx = ZF_BLOCK_MAX + 1;
            c->dstMaxHint = (size_t)mx; c->dstSlotsHint = (size_t)slotsWanted; }
        c->hostPipe = true;
        r = zhip_decompress_batch_device(c, c->hSrc.p, dSegs + lo, cnt, c->hDst.p, dSegs + n + lo, dSizes + lo, dStatus + lo, c->hpCompute);
        c->hostPipe = false; c->dstMaxHint = 0; c->dstSlotsHint = 0;
        if (r) return fail(r);
        if (hipEventRecord(evK, c->hpCompute) != hipSuccess || hipStreamWaitEvent(c->hpD2H, evK, 0) != hipSuccess) return fail(ZHIP_ERR_HIP);
        // the chunk's output goes straight into its (pinned) result payload
        ob[k].data = pin_pool().take((size_t)outBytes);
        ob[k].segs = (zhip_segment*)malloc((cnt ? cnt : 1) * sizeof(zhip_segment));
        if (!ob[k].data || !ob[k].segs) return fail(ZHIP_ERR_NO_MEMORY, lo);
        ob[k].dataSize = (size_t)outBytes; ob[k].nSegs = cnt;
        if ((outBytes && hipMemcpyAsync(ob[k].data, (uint8_t*)c->hDst.p + segs[n + lo].offset, outBytes, hipMemcpyDeviceToHost, c->hpD2H) != hipSuccess) ||
            hipMemcpyAsync(hSizes + lo, dSizes + lo, cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, c->hpD2H) != hipSuccess ||
            hipMemcpyAsync(hStatus + lo, dStatus + lo, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->hpD2H) != hipSuccess) return fail(ZHIP_ERR_HIP);
    }
    if (hipStreamSynchronize(c->hpD2H) != hipSuccess || hipStreamSynchronize(c->hpCompute) != hipSuccess) return fail(ZHIP_ERR_HIP);
    for (size_t k = 0; k < nChunks; k++) {
        const size_t lo = cut[k], hi = cut[k + 1];
        const uint64_t base = segs[n + lo].offset;
        for (size_t i = lo; i < hi; i++) {
            if (hStatus[i]) return fail(ZHIP_ERR_ZSTD, i, hStatus[i]);
            if (hSizes[i] != segs[n + i].length && !(allowShort && hSizes[i] < segs[n + i].length)) return fail(ZHIP_ERR_SIZE_MISMATCH, i, 0, hSizes[i], segs[n + i].length);
            ob[k].segs[i - lo].offset = segs[n + i].offset - base; ob[k].segs[i - lo].length = hSizes[i];
        }
    }
    (void)hipEventDestroy(evUp); (void)hipEventDestroy(evK);
    if (nChunks == 0) empty_outbuf(&ob[0]);
    *out = ob; *nOut = nChunks ? nChunks : 1;
    tls_trim(c);
    return ZHIP_ERR_NONE;
}

static int compress_batch_one(const zhip_cparams* params, const zhip_item* items, size_t n, zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    if (err) memset(err, 0, sizeof *err);
    *out = nullptr; *nOut = 0;
    zhip_ctx* c = tls_ctx();
    if (!c) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    zhip_cparams defaults; memset(&defaults, 0, sizeof defaults);
    defaults.level = 3; defaults.contentSizeFlag = 1; defaults.dictIDFlag = 1;
    int r = zhip_ctx_set_cparams(c, params ? params : &defaults);
    if (r < 0) return set_err(err, ZHIP_ERR_ZSTD, 0, -r);
    if (r) return set_err(err, r, 0, 0);
    if (n == 0) {                                     // nothing to compress: the parameters / dictionary have been digested (what precompute_compress asks for)
        zhip_outbuf* e = (zhip_outbuf*)calloc(1, sizeof(zhip_outbuf));
        if (!e) return set_err(err, ZHIP_ERR_NO_MEMORY, 0, 0);
        empty_outbuf(e); *out = e; *nOut = 1;
        return ZHIP_ERR_NONE;
    }
    // like compress_worker (compressor.c:913-947) every item gets a ZSTD_compressBound-sized slot; the frames are compacted on the
    // device, chunk by chunk, and only they travel back
    std::vector<zhip_segment> segs(2 * n);
    uint64_t srcTotal = 0, dstTotal = 0;
    for (size_t i = 0; i < n; i++) {
        if (items[i].srcSize >= (1ull << 31)) {
            g_lastError = "inputs of 2 GiB and more are not implemented in the HIP backend";
            return set_err(err, ZHIP_ERR_UNSUPPORTED, i, 0);
        }
        segs[i].offset = srcTotal; segs[i].length = items[i].srcSize; srcTotal += items[i].srcSize;
        uint64_t b = zhip_compress_bound(items[i].srcSize);
        b = (b + 15) & ~(uint64_t)15;
        segs[n + i].offset = dstTotal; segs[n + i].length = b; dstTotal += b;
    }
    if (host_pipe_init(c)) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    // the match kernel is a per-frame latency chain (its time barely depends on the batch below ~16 K frames), so compress chunks are
    // large: two of them overlap one's upload with the other's kernels, more would only add chains end to end
    // (round 4 ran these chunks side by side on their own streams, each in its own part of the encode arenas -- +8 % at best and an unexplained 5 x
    // slow mode in three of eight configurations; removed in round 5, git tag r04-experiments has it. The ceiling of this shape is the search:
    // 65 536 frames are 410-480 ms of transaction-bound work however they are cut, with the first chunk's upload in front and entropy coding,
    // compaction and the copy back behind: DESIGN.md section 3.)
    const std::vector<size_t> cut = host_chunks(segs.data(), n, (uint64_t)4 << 30, c->knob.hchunkE, n > c->knob.hchunkE ? c->knob.hchunkE0 : 0);
    const size_t nChunks = cut.size() - 1;
    // device: sources, slots, dense frames, segment table, [sizes | offsets | chunk totals | status]
    const size_t metaBytes = n * (2 * sizeof(uint64_t) + sizeof(int32_t)) + (nChunks + 1) * sizeof(uint64_t) + 32;
    if (c->hSrc.reserve(srcTotal + 16) || c->hDst.reserve(dstTotal + 16) || c->hDense.reserve(dstTotal + 16) || c->hSegs.reserve(2 * n * sizeof(zhip_segment) + 16) ||
        c->hStatus.reserve(metaBytes)) return set_err(err, g_reserveRc, 0, 0);
    if (ensure_pinned(c, metaBytes)) return set_err(err, ZHIP_ERR_HIP, 0, 0);
    zhip_outbuf* ob = (zhip_outbuf*)calloc(nChunks ? nChunks : 1, sizeof(zhip_outbuf));
    if (!ob) return set_err(err, ZHIP_ERR_NO_MEMORY, 0, 0);
    std::vector<hipEvent_t> evMeta(nChunks, nullptr);
    hipEvent_t evUp = nullptr;
    auto fail = [&](int kind, size_t index = 0, int zerr = 0) -> int {
        if (kind == ZHIP_ERR_HIP) hip_fail(hipGetLastError(), "host compress pipeline");
        (void)hipDeviceSynchronize();
        for (auto e : evMeta) if (e) (void)hipEventDestroy(e);
        if (evUp) (void)hipEventDestroy(evUp);
        zhip_free_outbufs(ob, nChunks ? nChunks : 1, 1);
        return set_err(err, kind, index, zerr);
    };
    if (hipEventCreateWithFlags(&evUp, hipEventDisableTiming) != hipSuccess) return fail(ZHIP_ERR_HIP);
    if (hipMemcpyAsync(c->hSegs.p, segs.data(), 2 * n * sizeof(zhip_segment), hipMemcpyHostToDevice, c->hpH2D) != hipSuccess ||
        hipStreamSynchronize(c->hpH2D) != hipSuccess) return fail(ZHIP_ERR_HIP);
    uint64_t* dSizes = (uint64_t*)c->hStatus.p; uint64_t* dOffs = dSizes + n; uint64_t* dTotals = dOffs + n; int32_t* dStatus = (int32_t*)(dTotals + nChunks + 1);
    uint64_t* hSizes = (uint64_t*)c->pinned; uint64_t* hTotals = hSizes + 2 * n; int32_t* hStatus = (int32_t*)(hTotals + nChunks + 1);
    const zhip_segment* dSegs = (const zhip_segment*)c->hSegs.p;
    // the host learns a chunk's total one chunk behind: while chunk k is being packed / copied / compressed, chunk k-1's sizes have
    // arrived, its payload is allocated (exact size) and its frames travel back on the third stream
    auto collect = [&](size_t k) -> int {
        const size_t lo = cut[k], hi = cut[k + 1], cnt = hi - lo;
        if (hipEventSynchronize(evMeta[k]) != hipSuccess) return fail(ZHIP_ERR_HIP);
        (void)hipEventDestroy(evMeta[k]); evMeta[k] = nullptr;
        for (size_t i = lo; i < hi; i++) if (hStatus[i]) return fail(ZHIP_ERR_ZSTD, i, hStatus[i]);
        const uint64_t total = hTotals[k];
        ob[k].data = pin_pool().take((size_t)total);
        ob[k].segs = (zhip_segment*)malloc((cnt ? cnt : 1) * sizeof(zhip_segment));
        if (!ob[k].data || !ob[k].segs) return fail(ZHIP_ERR_NO_MEMORY, lo);
        ob[k].dataSize = (size_t)total; ob[k].nSegs = cnt;
        uint64_t o = 0;
        for (size_t i = lo; i < hi; i++) { ob[k].segs[i - lo].offset = o; ob[k].segs[i - lo].length = hSizes[i]; o += hSizes[i]; }
        // the compaction finished before evMeta (same stream), so the copy needs no further dependency
        if (total && hipMemcpyAsync(ob[k].data, (uint8_t*)c->hDense.p + segs[n + lo].offset, total, hipMemcpyDeviceToHost, c->hpD2H) != hipSuccess) return fail(ZHIP_ERR_HIP);
        return 0;
    };
    for (size_t k = 0; k < nChunks; k++) {
        const size_t lo = cut[k], hi = cut[k + 1], cnt = hi - lo;
        hipStream_t sk = c->hpCompute;
        if (upload_items(c, items, segs.data(), lo, hi, evUp)) return fail(ZHIP_ERR_HIP);
        if (hipStreamWaitEvent(sk, evUp, 0) != hipSuccess) return fail(ZHIP_ERR_HIP);
        { size_t mx = 0; for (size_t i = lo; i < hi; i++) if (items[i].srcSize > mx) mx = items[i].srcSize; c->srcMaxHint = mx; }
        r = zhip_compress_batch_device(c, c->hSrc.p, dSegs + lo, cnt, c->hDst.p, dSegs + n + lo, dSizes + lo, dStatus + lo, sk);
        c->srcMaxHint = 0;
        if (r) return fail(r);
        hipLaunchKernelGGL(zhip_scan_sizes_kernel, dim3(1), dim3(1024), 0, sk, dSizes + lo, dStatus + lo, (uint32_t)cnt, dOffs + lo, dTotals + k);
        const uint32_t gridC = (uint32_t)(cnt < (size_t)c->numCU * 16 ? cnt : (size_t)c->numCU * 16);
        hipLaunchKernelGGL(zhip_compact_kernel, dim3(gridC ? gridC : 1), dim3(64), 0, sk, (const uint8_t*)c->hDst.p, dSegs + n + lo, dSizes + lo, dStatus + lo,
                           dOffs + lo, (uint32_t)cnt, (uint8_t*)c->hDense.p + segs[n + lo].offset);
        // sizes, status and the chunk total ride the compute stream (a few hundred KiB) so that nothing queues behind a later chunk's kernels
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(hSizes + lo, dSizes + lo, cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, sk) != hipSuccess ||
            hipMemcpyAsync(hStatus + lo, dStatus + lo, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, sk) != hipSuccess ||
            hipMemcpyAsync(hTotals + k, dTotals + k, sizeof(uint64_t), hipMemcpyDeviceToHost, sk) != hipSuccess ||
            hipEventCreateWithFlags(&evMeta[k], hipEventDisableTiming) != hipSuccess || hipEventRecord(evMeta[k], sk) != hipSuccess) return fail(ZHIP_ERR_HIP);
        if (k >= 1) { const int e = collect(k - 1); if (e) return e; }
    }
    if (nChunks) { const int e = collect(nChunks - 1); if (e) return e; }
    if (hipStreamSynchronize(c->hpD2H) != hipSuccess || hipStreamSynchronize(c->hpCompute) != hipSuccess) return fail(ZHIP_ERR_HIP);
    (void)hipEventDestroy(evUp);
    if (nChunks == 0) empty_outbuf(&ob[0]);
    *out = ob; *nOut = nChunks ? nChunks : 1;
    tls_trim(c);
    return ZHIP_ERR_NONE;
}


// ------------------------------------------------------------------------------------------ one call, every device (round 6)
// The reference fans a batch out INSIDE multi_compress_to_buffer / multi_decompress_to_buffer: bytesPerWorker = total / workers, a contiguous run of
// items per worker, one destination buffer set per worker, the collection in worker order, the first failing item reported (c-ext/compressor.c:1127-1298,
// c-ext/decompressor.c:1237-1455). Here a worker is a DEVICE: zhip_compress_batch / zhip_decompress_batch cut the item list by the same rule, hand every
// run to a host thread bound to its device (one persistent thread + one context + its own staging per device slot; the pinned payload pool is the process's),
// and return the devices' zhip_outbufs in device order -- which is item order. No collective: the collection holds one BufferWithSegments per chunk of each
// device. ZHIP_DEVICES=0,1,... selects the device slots (a device may be listed twice: two contexts on one GPU -- how the split is tested on a
// one-GPU box); unset = every visible device, but never more devices than ZHIP_DEVICE_MIN_BYTES (default 256 MiB) of input each -- small batches stay on the
// calling thread's current device, exactly as before.
extern "C" size_t zhip_partition_by_bytes(const uint64_t* sizes, size_t n, size_t workers, size_t* bounds)
{
    // compressor.c:1127-1216: never more workers than items (:1151); walk the items, close a worker's run once its bytes reach total / workers; the
    // last worker takes the rest. bounds[2w], bounds[2w+1] = [start, end) of worker w; returns the workers that got a run.
    if (workers < 1) workers = 1;
    if (n && workers > n) workers = n;
    if (!n) return 0;
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += sizes[i];
    const uint64_t per = total / workers;
    size_t w = 0, start = 0; uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        acc += sizes[i];
        if (w == workers - 1) continue;
        if (acc >= per) { bounds[2 * w] = start; bounds[2 * w + 1] = i + 1; w++; start = i + 1; acc = 0; }
    }
    if (start < n) { bounds[2 * w] = start; bounds[2 * w + 1] = n; w++; }
    return w;
}

namespace {
struct DevJob { std::function<void()> fn; };
struct DevWorker {
    int device = 0;
    std::mutex mu; std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::thread th;
    void loop()
    {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<void()> fn;
            { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !q.empty(); }); fn = std::move(q.front()); q.pop_front(); }
            fn();
        }
    }
    void post(std::function<void()> fn) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(fn)); } cv.notify_one(); }
};
struct DevPool {
    std::mutex mu;
    bool parsed = false, explicitList = false;
    std::vector<int> slots;                      // device ordinal of every slot
    std::vector<DevWorker*> workers;             // created on first use, never destroyed (their contexts live as long as the process)
    uint64_t minBytes = (uint64_t)256 << 20;
    void parse()
    {
        if (parsed) return;
        parsed = true;
        if (const char* e = getenv("ZHIP_DEVICE_MIN_BYTES")) minBytes = strtoull(e, nullptr, 10);
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
        if (const char* e = getenv("ZHIP_DEVICES")) {
            explicitList = true;
            for (const char* p = e; *p;) {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                if (v >= 0 && v < count) slots.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
        } else for (int d = 0; d < count; d++) slots.push_back(d);
        workers.assign(slots.size(), nullptr);
        if (slots.size() > 1) g_pinPortable.store(true);       // payload blocks serve any device's copy engines from the first one on
    }
    DevWorker* worker(size_t slot)
    {
        std::lock_guard<std::mutex> l(mu);
        if (!workers[slot]) { DevWorker* w = new DevWorker(); w->device = slots[slot]; w->th = std::thread([w] { w->loop(); }); w->th.detach(); workers[slot] = w; }
        return workers[slot];
    }
    // device slots a batch of `n` items / `bytes` input bytes is cut over (0: the calling thread's current device, inline)
    size_t slotsFor(size_t n, uint64_t bytes)
    {
        { std::lock_guard<std::mutex> l(mu); parse(); }
        size_t d = slots.size();
        if (!explicitList) {
            if (d <= 1) return 0;
            const uint64_t byBytes = minBytes ? bytes / minBytes : d;
            if (byBytes < d) d = (size_t)byBytes;
            if (d <= 1) return 0;
        }
        if (d > n) d = n;
        return d;
    }
};
DevPool& dev_pool() { static DevPool* p = new DevPool(); return *p; }

struct DevResult { int rc = 0; zhip_error err; zhip_outbuf* out = nullptr; size_t nOut = 0; std::string text; bool done = false; };

// runs fn(slot, lo, hi, &result) for every run of the partition on its device's thread; gathers the devices' buffers in order
template <typename F>
int fan_out(size_t d, const std::vector<uint64_t>& sizes, size_t n, F&& one, zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    std::vector<size_t> bounds(2 * d);
    const size_t used = zhip_partition_by_bytes(sizes.data(), n, d, bounds.data());
    std::vector<DevResult> res(used);
    std::mutex mu; std::condition_variable cv; size_t pending = used;
    for (size_t w = 0; w < used; w++) {
        const size_t lo = bounds[2 * w], hi = bounds[2 * w + 1];
        dev_pool().worker(w)->post([&, w, lo, hi] {
            DevResult& r = res[w];
            memset(&r.err, 0, sizeof r.err);
            r.rc = one(lo, hi, &r.out, &r.nOut, &r.err);
            if (r.rc) r.text = g_lastError;
            if (w < 64 && g_tls.c) g_slotBytes[w].store(g_tls.c->device_bytes());
            { std::lock_guard<std::mutex> l(mu); r.done = true; pending--; }
            cv.notify_one();
        });
    }
    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return pending == 0; }); }
    // the first failing item (the lowest index: runs are in item order), as the reference reports it (compressor.c:1225-1253)
    for (size_t w = 0; w < used; w++) if (res[w].rc) {
        for (size_t v = 0; v < used; v++) if (!res[v].rc) zhip_free_outbufs(res[v].out, res[v].nOut, 1);
        if (err) { *err = res[w].err; err->index += bounds[2 * w]; }
        g_lastError = res[w].text;
        return res[w].rc;
    }
    size_t total = 0;
    for (size_t w = 0; w < used; w++) total += res[w].nOut;
    zhip_outbuf* all = (zhip_outbuf*)calloc(total ? total : 1, sizeof(zhip_outbuf));
    if (!all) { for (size_t w = 0; w < used; w++) zhip_free_outbufs(res[w].out, res[w].nOut, 1); if (err) { memset(err, 0, sizeof *err); err->kind = ZHIP_ERR_NO_MEMORY; } return ZHIP_ERR_NO_MEMORY; }
    size_t k = 0;
    for (size_t w = 0; w < used; w++) { for (size_t j = 0; j < res[w].nOut; j++) all[k++] = res[w].out[j]; free(res[w].out); }
    *out = all; *nOut = total;
    return ZHIP_ERR_NONE;
}
}

extern "C" int zhip_batch_devices(int* devices, int cap)
{
    DevPool& p = dev_pool();
    { std::lock_guard<std::mutex> l(p.mu); p.parse(); }
    for (int i = 0; i < cap && i < (int)p.slots.size(); i++) devices[i] = p.slots[i];
    return (int)p.slots.size();
}

extern "C" int zhip_compress_batch(const zhip_cparams* params, const zhip_item* items, size_t n, zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    uint64_t bytes = 0;
    for (size_t i = 0; i < n; i++) bytes += items[i].srcSize;
    const size_t d = dev_pool().slotsFor(n, bytes);
    if (d == 0) return compress_batch_one(params, items, n, out, nOut, err);
    if (err) memset(err, 0, sizeof *err);
    *out = nullptr; *nOut = 0;
    std::vector<uint64_t> sizes(n);
    for (size_t i = 0; i < n; i++) sizes[i] = items[i].srcSize;
    return fan_out(d, sizes, n, [&](size_t lo, size_t hi, zhip_outbuf** o, size_t* no, zhip_error* e) { return compress_batch_one(params, items + lo, hi - lo, o, no, e); }, out, nOut, err);
}

extern "C" int zhip_decompress_batch(const zhip_dparams* params, const zhip_item* items, size_t n, int requireSizes,
                                     zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    // the reference's decompress dispatcher balances by COMPRESSED bytes (decompressor.c:1237-1320: framePointers' sourceSize)
    uint64_t bytes = 0, outBytes = 0;
    for (size_t i = 0; i < n; i++) { bytes += items[i].srcSize; outBytes += items[i].dstSize; }
    const size_t d = dev_pool().slotsFor(n, outBytes > bytes ? outBytes : bytes);
    if (d == 0) return decompress_batch_one(params, items, n, requireSizes, out, nOut, err);
    if (err) memset(err, 0, sizeof *err);
    *out = nullptr; *nOut = 0;
    std::vector<uint64_t> sizes(n);
    for (size_t i = 0; i < n; i++) sizes[i] = items[i].srcSize;
    return fan_out(d, sizes, n, [&](size_t lo, size_t hi, zhip_outbuf** o, size_t* no, zhip_error* e) { return decompress_batch_one(params, items + lo, hi - lo, requireSizes, o, no, e); }, out, nOut, err);
}
