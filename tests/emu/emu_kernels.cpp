#include <vector>
// tests/emu/emu_kernels.cpp -- compiles the PRODUCT kernel sources for the host wave emulator (ZHIP_EMU).
// Test infrastructure only: lets tests/test_emu_*.py exercise kernel logic without a GPU. Never shipped.
#define ZHIP_EMU 1
extern "C" { long zd_trace_pos = -1; long zd_cur_frame = -1; long zd_stat[16]; }
#include "../../python-zstandard_amd/csrc/zhip_decode_pipeline.hpp"
#include "../../python-zstandard_amd/csrc/zhip_encode_kernel.hpp"
#include "../../python-zstandard_amd/csrc/zhip_cparams.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

static ZdLDS g_lds;
static uint32_t g_magicless = 0;     // frame format of the next emulated launches (1: ZSTD_f_zstd1_magicless)
struct DecLaunch { const ZhipDecodeArgs* a; };
static void dec_lane(void* p) { zd_kernel_body(*((DecLaunch*)p)->a, g_lds); }

extern "C" int emu_decompress_batch(const uint8_t* src, const uint64_t* srcSegs, uint32_t n, uint8_t* dst,
                                    const uint64_t* dstSegs, uint64_t* outSizes, int32_t* status,
                                    const uint8_t* dictContent, uint32_t dictContentSize, uint32_t dictID,
                                    const ZhipDictEntropy* de, uint32_t nBlocks)
{
    ZhipDecodeArgs a; memset(&a, 0, sizeof(a));
    uint32_t counter = 0;
    a.src = src; a.srcSegs = srcSegs; a.dst = dst; a.dstSegs = dstSegs; a.outSizes = outSizes; a.status = status;
    a.scratch = (uint8_t*)malloc((size_t)nBlocks * ZHIP_LIT_STRIDE);
    a.counter = &counter; a.n = n; a.dictID = dictID; a.dictContent = dictContent; a.dictContentSize = dictContentSize;
    a.dictEntropy = de; a.maxWindowSize = (1ull << 27) + 1; a.magicless = g_magicless;
    memset(&g_lds, 0xA5, sizeof g_lds);
    DecLaunch l = { &a };
    zhemu::run_grid(nBlocks, dec_lane, &l);
    free(a.scratch);
    return 0;
}

struct DictLaunch { const uint8_t* dict; uint32_t size; ZhipDictEntropy* de; };
static void dict_lane(void* p) { DictLaunch* l = (DictLaunch*)p; zd_dict_body(l->dict, l->size, l->de, g_lds); }
extern "C" int emu_parse_dict(const uint8_t* dict, uint32_t size, ZhipDictEntropy* de)
{
    memset(de, 0, sizeof *de);
    DictLaunch l = { dict, size, de };
    zhemu::run_grid(1, dict_lane, &l);
    return de->status;
}
extern "C" uint32_t emu_dict_entropy_size(void) { return (uint32_t)sizeof(ZhipDictEntropy); }

static ZeLDS g_elds;
// compression dictionary digested by the product's own kernels under emulation (mirrors zhip_ctx_set_cparams)
static std::vector<uint8_t> g_cdBlob; static ZhipDictEntropy g_cdEntropy; static ZeCDict g_cd; static std::vector<uint32_t> g_cdTables;
static bool g_hasCD = false;
// explicit compression parameters of the next emulated launches (all zero: derive everything from the level) and the frame format
static zhip_compression_parameters g_ov;
extern "C" void emu_set_cparams(uint32_t w, uint32_t c, uint32_t h, uint32_t s, uint32_t mm, uint32_t tl, int32_t strat, uint32_t magicless)
{
    g_ov.windowLog = w; g_ov.chainLog = c; g_ov.hashLog = h; g_ov.searchLog = s; g_ov.minMatch = mm; g_ov.targetLength = tl; g_ov.strategy = strat;
    g_magicless = magicless;
}
static uint32_t emu_table_stride(const ZeRows& rows)            // mirrors zhip_compress_batch_device
{
    uint32_t stride = 0;
    for (int t = 2; t < 4; t++) {
        const int32_t* r = rows.r[t];
        if (r[6] != 1 && r[6] != 2) continue;
        const int w = r[0] < (t == 2 ? 17 : 14) ? r[0] : (t == 2 ? 17 : 14);
        const int h = r[2] > w + 1 ? w + 1 : r[2], cl = r[1] > w ? w : r[1];
        const uint32_t bytes = (4u << h) + (r[6] == 2 ? (4u << cl) : 0u);
        if (bytes > stride) stride = bytes;
    }
    if (stride < (4u << 10)) stride = 4u << 10;
    if (stride > (12u << 17)) stride = 12u << 17;
    return stride;
}
struct CDLaunch { ZeRows rows; };
static void cdict_lane(void* p)
{
    const size_t cells = (size_t)1 << ZE_CDICT_MAX_HLOG;
    ze_cdict_body(g_cdBlob.data(), (uint32_t)(g_cdBlob.size() - 16), &g_cdEntropy, ((CDLaunch*)p)->rows, &g_cd,
                  g_cdTables.data(), g_cdTables.data() + cells, g_cdTables.data() + 2 * cells, g_elds);
}
extern "C" int emu_set_cdict(const uint8_t* dict, uint32_t size, int level)
{
    g_hasCD = false;
    if (!dict || !size) return 0;
    g_cdBlob.assign(dict, dict + size); g_cdBlob.resize(size + 16, 0);
    memset(&g_cdEntropy, 0, sizeof g_cdEntropy); memset(&g_cd, 0, sizeof g_cd);
    DictLaunch l = { g_cdBlob.data(), size, &g_cdEntropy };
    zhemu::run_grid(1, dict_lane, &l);
    if (g_cdEntropy.status) return g_cdEntropy.status;
    g_cdTables.assign(3 * ((size_t)1 << ZE_CDICT_MAX_HLOG), 0xDEADBEEFu);
    CDLaunch c; zh_resolve_rows(&c.rows, level, &g_ov);
    zhemu::run_grid(1, cdict_lane, &c);
    if (g_cd.status) return g_cd.status;
    g_hasCD = true;
    return 0;
}
static void attach_cdict(ZhipEncodeArgs& a)
{
    if (!g_hasCD) return;
    a.cdict = &g_cd;
    a.cdictContent = g_cdBlob.data() + (g_cdEntropy.hufCount ? g_cdEntropy.contentOffset : 0u);
    a.cdictHashLong = g_cdTables.data(); a.cdictHashSmall = g_cdTables.data() + ((size_t)1 << ZE_CDICT_MAX_HLOG);
}
struct EncLaunch { const ZhipEncodeArgs* a; };
static ZeLDSMulti g_eldsm;
static void enc_lane(void* p) { ze_kernel_body(*((EncLaunch*)p)->a, g_elds, g_eldsm); }
extern "C" int emu_compress_batch(const uint8_t* src, const uint64_t* srcSegs, uint32_t n, uint8_t* dst, const uint64_t* dstSegs,
                                  uint64_t* outSizes, int32_t* status, int level, uint32_t flags, uint32_t nBlocks)
{
    ZhipEncodeArgs a; memset(&a, 0, sizeof(a));
    uint32_t counter = 0;
    a.src = src; a.srcSegs = srcSegs; a.dst = dst; a.dstSegs = dstSegs; a.outSizes = outSizes; a.status = status;
    a.workspace = (uint8_t*)malloc((size_t)nBlocks * ZHIP_ENC_STRIDE);
    a.counter = &counter; a.n = n; a.level = level; zh_resolve_rows(&a.rows, level, &g_ov); a.magicless = g_magicless;
    a.contentSizeFlag = flags & 1; a.checksumFlag = (flags >> 1) & 1; a.dictIDFlag = (flags >> 2) & 1;
    attach_cdict(a);
    memset(&g_elds, 0xA5, sizeof g_elds);
    EncLaunch l = { &a };
    zhemu::run_grid(nBlocks, enc_lane, &l);
    free(a.workspace);
    return 0;
}

// ---- phase-split decode pipeline under emulation
static ZpExecLDS g_xlds;
static ZpBinLDS g_binlds;
static uint32_t g_llBase[36], g_mlBase[53]; static uint8_t g_llBits[36], g_mlBits[56];
static uint32_t g_k1Lanes = 1;                   // dictionary batches: K1's lane-per-frame pass first (zp_lit_lanes_body), K1 over what it listed (0: a wave per frame, rounds 1-5)
static void k1lanes_lane(void* p) { zp_lit_lanes_body(*(const ZhipPipeArgs*)p); }
extern "C" void emu_set_k1_lanes(uint32_t v) { g_k1Lanes = v; }
static void k1_lane(void* p) { zp_lit_body(*(const ZhipPipeArgs*)p, g_lds); }
static ZpPreLDS g_prelds;
static uint32_t g_k0 = 1;                        // K0 (zp_pre_body: a lane per frame walks K1's serial descriptions) before K1 -- as zhip_decompress_batch_device runs it; 0: K1 parses everything itself
static void k0_lane(void* p) { zp_pre_body(*(const ZhipPipeArgs*)p, g_prelds); }
extern "C" void emu_set_k0(uint32_t v) { g_k0 = v; }
static void kb_lane(void* p) { zp_bin_body(*(const ZhipPipeArgs*)p, g_binlds); }
static ZpHufKernelLDS g_huflds;
static void kh_lane(void* p) { zp_huf_body(*(const ZhipPipeArgs*)p, g_huflds); }
static ZpSeqQLDS g_seqqlds;
static void k2_lane(void* p) { const ZhipPipeArgs& a = *(const ZhipPipeArgs*)p; if (a.itemCap) zp_seqq_body<true>(a, g_seqqlds); else zp_seqq_body<false>(a, g_seqqlds); }
static void k3_lane(void* p)
{
    const ZhipPipeArgs& a = *(const ZhipPipeArgs*)p;
    if (a.itemCap) { if (a.dictContent) zp_exec_body<true, false, true>(a, g_xlds); else zp_exec_body<false, false, true>(a, g_xlds); }
    else if (a.dictContent) zp_exec_body<true, false>(a, g_xlds); else zp_exec_body<false, false>(a, g_xlds);
}
static void kx_lane(void* p) { zp_check_body(*(const ZhipPipeArgs*)p); }
static uint32_t g_ckLater = 1;                  // content checksums verified by KX after K3 (a lane per frame), as the product does; 0: by K1 / K3 on one lane (rounds 1-5)
extern "C" void emu_set_check_later(uint32_t v) { g_ckLater = v; }
static void k1mb_lane(void* p) { zp_lit_mb_body(*(const ZhipPipeArgs*)p, g_lds); }
static uint64_t g_arenaBudget16 = 0;                    // compact decode arena: != 0 overrides the harness' worst-case budget, in 16-byte units (what runs out is the generic kernel's)
extern "C" void emu_set_arena_budget(uint64_t units16) { g_arenaBudget16 = units16; }
static uint32_t g_mbPerFrame = 0;              // several-block mode of the pipeline harness: item slots per frame (0 = off), mirrors zhip_decompress_batch_device
extern "C" void emu_set_blocks(uint32_t perFrame) { g_mbPerFrame = perFrame; }
// decompression dictionary for the pipeline harness (mirrors zhip_ctx_set_ddict): blob, parsed entropy section, ready-made tables
static std::vector<uint8_t> g_ddBlob; static ZhipDictEntropy g_ddEntropy; static ZhipDictTables g_ddTables; static bool g_ddHas = false, g_ddEnt = false;
struct DTLaunch { const ZhipDictEntropy* de; ZhipDictTables* out; };
static void dt_lane(void* p) { DTLaunch* l = (DTLaunch*)p; zp_dict_tables_body(l->de, l->out, g_lds); }
extern "C" int emu_set_ddict(const uint8_t* dict, uint32_t size, int rawContent)
{
    g_ddHas = g_ddEnt = false;
    if (!dict || !size) return 0;
    g_ddBlob.assign(dict, dict + size); g_ddBlob.resize(size + 64, 0);
    memset(&g_ddEntropy, 0, sizeof g_ddEntropy); memset(&g_ddTables, 0, sizeof g_ddTables);
    if (!rawContent && size >= 8 && zh_ld32(dict) == 0xEC30A437u) {
        memset(&g_lds, 0xA5, sizeof g_lds);
        DictLaunch l = { g_ddBlob.data(), size, &g_ddEntropy };
        zhemu::run_grid(1, dict_lane, &l);
        if (g_ddEntropy.status) return g_ddEntropy.status;
        if (g_ddEntropy.hufCount) {
            memset(&g_lds, 0xA5, sizeof g_lds);
            DTLaunch t = { &g_ddEntropy, &g_ddTables };
            zhemu::run_grid(1, dt_lane, &t);
            if (g_ddTables.status) return g_ddTables.status;
            g_ddEnt = true;
        }
    }
    g_ddHas = true;
    return 0;
}
extern "C" int emu_decompress_pipeline(const uint8_t* src, const uint64_t* srcSegs, uint32_t n, uint8_t* dst,
                                       const uint64_t* dstSegs, uint64_t* outSizes, int32_t* status, uint32_t nBlocks, uint32_t chunk)
{
    ZhipPipeArgs a; memset(&a, 0, sizeof(a));
    static uint32_t counters[ZP_CNT_WORDS + 1]; memset(counters, 0, sizeof counters);      // the slot's words (incl. the work orders' bin counters), then the fallback length
    if (chunk == 0 || chunk > n) chunk = n ? n : 1;
    const bool mb = g_mbPerFrame != 0;
    const size_t slots = (size_t)chunk * (mb ? g_mbPerFrame : 1u);
    a.src = src; a.srcSegs = srcSegs; a.dst = dst; a.dstSegs = dstSegs; a.outSizes = outSizes; a.status = status;
    a.meta = (ZdMeta*)calloc(slots, sizeof(ZdMeta));
    // (ONE compact arena for literals and sequences with a budget -- worst case by default: every frame a full literal slot and its K2 group's longest possible room)
    const size_t arenaBudget16 = g_arenaBudget16 ? (size_t)g_arenaBudget16 : slots * ((ZP_SEQ_STRIDE * 2 + ZP_LIT_STRIDE) / 16 + 1);
    uint8_t* const litAlloc = (uint8_t*)malloc(arenaBudget16 * 16 + 512 + ZP_LIT_FRONT); a.litArena = litAlloc + ZP_LIT_FRONT; a.seqArena = (uint64_t*)a.litArena;
    a.bases = (uint32_t*)malloc(slots * 8); memset(a.bases, 0xA5, slots * 8); a.arenaBudget16 = (uint32_t)arenaBudget16;
    a.fseTables = (uint16_t*)malloc(slots * ZP_FSE_CELLS * 2);
    a.order = (uint32_t*)calloc(slots, 4);
    a.hufTables = (uint16_t*)malloc(slots * ZP_HUF_CELLS * 2 + 64);
    a.orderLit = (uint32_t*)calloc(slots, 4);
    if (mb) { a.itemFrame = (uint32_t*)malloc(slots * 4); a.itemReps = (uint32_t*)malloc(slots * 16); a.frameRecs = (ZpFrameRec*)malloc((size_t)chunk * sizeof(ZpFrameRec));
              memset(a.itemFrame, 0xA5, slots * 4); memset(a.itemReps, 0xA5, slots * 16); memset(a.frameRecs, 0xA5, (size_t)chunk * sizeof(ZpFrameRec)); }
    a.counters = counters; a.fallbackCount = &counters[ZP_CNT_WORDS]; a.fallbackList = (uint32_t*)calloc(n ? n : 1, 4);
    a.maxWindowSize = (1ull << 27) + 1; a.magicless = g_magicless;
    if (g_ddHas) {
        const uint32_t co = g_ddEnt ? g_ddEntropy.contentOffset : 0u;
        a.dictID = g_ddEnt ? g_ddEntropy.dictID : 0u; a.dictContent = g_ddBlob.data() + co; a.dictContentSize = (uint32_t)(g_ddBlob.size() - 64) - co;
        a.dictEntropy = g_ddEnt ? &g_ddEntropy : nullptr; a.dictTables = g_ddEnt ? &g_ddTables : nullptr;
    }
    a.k1Lanes = g_k1Lanes && a.dictEntropy ? 1u : 0u;       // (mirrors zhip_decompress_batch_device; cleared below in the several-block mode)
    ZpPre* pre = nullptr;
    a.ckLater = g_ckLater;
    for (uint32_t first = 0; first < n; first += chunk) {
        a.first = first; a.count = n - first < chunk ? n - first : chunk;
        for (uint32_t q = 0; q < ZP_CNT_WORDS; q++) counters[q] = 0;
        memset(&g_lds, 0xA5, sizeof g_lds); memset(&g_xlds, 0xA5, sizeof g_xlds); memset(&g_binlds, 0xA5, sizeof g_binlds);   // LDS is not zeroed on hardware
        a.itemCap = mb ? a.count * g_mbPerFrame : 0u;
        if (mb) a.k1Lanes = 0;
        if (a.k1Lanes) zhemu::run_grid(nBlocks, k1lanes_lane, &a);
        a.pre = nullptr;
        if (g_k0 && !mb && !a.dictEntropy) {          // (mirrors zhip_decompress_batch_device)
            if (!pre) pre = (ZpPre*)malloc(slots * sizeof(ZpPre));
            memset(pre, 0xA5, slots * sizeof(ZpPre)); memset(&g_prelds, 0xA5, sizeof g_prelds);
            a.pre = pre;
            zhemu::run_grid(nBlocks, k0_lane, &a);
        }
        zhemu::run_grid(nBlocks, mb ? k1mb_lane : k1_lane, &a);
        zhemu::run_grid(2 * (a.count < 8 ? 1u : 3u), kb_lane, &a);
        memset(&g_huflds, 0xA5, sizeof g_huflds);
        zhemu::run_grid(nBlocks, kh_lane, &a);
        memset(&g_seqqlds, 0xA5, sizeof g_seqqlds);
        zhemu::run_grid(nBlocks, k2_lane, &a);
        zhemu::run_grid(nBlocks, k3_lane, &a);
        if (a.ckLater) zhemu::run_grid(nBlocks, kx_lane, &a);
    }
    // generic kernel for everything the fast path declined
    ZhipDecodeArgs g; memset(&g, 0, sizeof(g));
    uint32_t counter = 0;
    g.src = src; g.srcSegs = srcSegs; g.dst = dst; g.dstSegs = dstSegs; g.outSizes = outSizes; g.status = status;
    g.scratch = (uint8_t*)malloc((size_t)nBlocks * ZHIP_LIT_STRIDE);
    g.counter = &counter; g.n = n; g.maxWindowSize = a.maxWindowSize; g.magicless = g_magicless; g.frameList = a.fallbackList; g.listCount = &counters[ZP_CNT_WORDS];
    g.dictID = a.dictID; g.dictContent = a.dictContent; g.dictContentSize = a.dictContentSize; g.dictEntropy = a.dictEntropy;
    DecLaunch l = { &g };
    zhemu::run_grid(nBlocks, dec_lane, &l);
    int nfb = (int)counters[ZP_CNT_WORDS];
    free(g.scratch); free(a.meta); free(litAlloc); free(a.fallbackList); free(a.fseTables); free(a.order); free(a.hufTables); free(a.orderLit);
    free(a.itemFrame); free(a.itemReps); free(a.frameRecs); free(a.bases); free(pre);
    return nfb;
}

// ---- two-kernel encoder under emulation
static void e1_lane(void* p) { ze_match_body(*(const ZhipEncodeArgs*)p); }
static void e2_lane(void* p) { ze_entropy_body(*(const ZhipEncodeArgs*)p, g_elds); }
static void ex_lane(void* p) { ze_trailer_body(*(const ZhipEncodeArgs*)p); }
static uint32_t g_xxLater = 1;                  // checksum trailers by EX after the entropy kernel (a lane per frame), as the product does; 0: by the entropy kernel on one lane
extern "C" void emu_set_trailer_later(uint32_t v) { g_xxLater = v; }
static uint32_t g_probes = 2;                   // probes per trip of the flat search (2, or 4: the latency-bound batches' form)
extern "C" void emu_set_probes(uint32_t v) { g_probes = v; }
static void e1f_lane(void* p) { if (g_probes == 4) ze_match_flat_body<4>(*(const ZhipEncodeArgs*)p); else if (g_probes == 3) ze_match_flat_body<3>(*(const ZhipEncodeArgs*)p); else ze_match_flat_body<2>(*(const ZhipEncodeArgs*)p); }
static void e1fmb_lane(void* p) { ze_match_flat_mb_body(*(const ZhipEncodeArgs*)p); }
static void split_lane(void* p) { ze_split_body(*(const ZhipEncodeArgs*)p, g_elds); }
static uint32_t g_mbCompress = 1;               // sources of several blocks in the flat match kernel: 0 off, 1 on, > 1 on with that many block slots per frame
extern "C" void emu_set_mb_compress(uint32_t v) { g_mbCompress = v; }
static uint32_t g_dictSlotMax = 0;              // != 0: ZhipEncodeArgs.slotSrcMax of dictionary batches (sources above it are the generic kernel's)
extern "C" void emu_set_dict_slot_max(uint32_t v) { g_dictSlotMax = v; }
static uint64_t g_mbHint = 0;                   // != 0: the several-block arenas are sized from this size HINT instead of the batch's largest source (a device-API caller's stale hint)
extern "C" void emu_set_mb_hint(uint64_t v) { g_mbHint = v; }
static uint32_t g_dictEpochs = 0, g_epoch = 0, g_epochShift = 0; static uint8_t* g_epochTables = nullptr; static size_t g_epochCap = 0;
extern "C" void emu_set_dict_epochs(uint32_t v) { g_dictEpochs = v; }        // 1: the flat dictionary search's tables carry launch numbers and persist between calls (the product's way); 0: zeroed by the kernel
static ZeSrcLDS<ZF_BLOCK_MAX> g_srclds;
static uint32_t g_e1LdsBytes = ZF_BLOCK_MAX;       // the LDS shape under emulation (the product picks it from the batch's largest source)
static void e1l_lane(void* p) { if (g_probes == 4) ze_match_lds_body<4>(*(const ZhipEncodeArgs*)p, g_srclds.b, g_e1LdsBytes); else ze_match_lds_body<2>(*(const ZhipEncodeArgs*)p, g_srclds.b, g_e1LdsBytes); }
extern "C" void emu_set_e1lds_bytes(uint32_t v) { g_e1LdsBytes = v; }
static uint32_t g_e1LdsMax = 0;                 // chunks of up to this many frames take the LDS-source match kernel (mirrors zhip_compress_batch_device's choice)
extern "C" void emu_set_e1lds_max(uint32_t v) { g_e1LdsMax = v; }
extern "C" int emu_compress_pipeline(const uint8_t* src, const uint64_t* srcSegs, uint32_t n, uint8_t* dst, const uint64_t* dstSegs,
                                     uint64_t* outSizes, int32_t* status, int level, uint32_t flags, uint32_t nBlocks, uint32_t chunk)
{
    ZhipEncodeArgs a; memset(&a, 0, sizeof(a));
    uint32_t counters[2] = {0, 0};
    if (chunk == 0 || chunk > n) chunk = n ? n : 1;
    a.src = src; a.srcSegs = srcSegs; a.dst = dst; a.dstSegs = dstSegs; a.outSizes = outSizes; a.status = status;
    a.workspace = (uint8_t*)malloc((size_t)nBlocks * ZHIP_ENC_STRIDE);
    a.counter = counters; a.n = n; a.level = level; zh_resolve_rows(&a.rows, level, &g_ov); a.magicless = g_magicless;
    a.contentSizeFlag = flags & 1; a.checksumFlag = (flags >> 1) & 1; a.dictIDFlag = (flags >> 2) & 1;
    free(a.workspace); a.workspace = (uint8_t*)malloc((size_t)nBlocks * ZE_E2_STRIDE + ZHIP_ENC_STRIDE);
    a.tableStride = emu_table_stride(a.rows);
    uint64_t maxSrc = 0; for (uint32_t i = 0; i < n; i++) if (srcSegs[2 * (size_t)i + 1] > maxSrc) maxSrc = srcSegs[2 * (size_t)i + 1];
    if (!g_hasCD && g_mbCompress && maxSrc > ZF_BLOCK_MAX && maxSrc < (1ull << ZE_MB_POS_BITS) - 8)          // mirrors zhip_compress_batch_device (mbcWanted)
        for (int t = 0; t < 2; t++) {
            const int32_t* r = a.rows.r[t];
            if (r[6] != 2) continue;
            int w = 17; while ((1ull << w) < maxSrc) w++;
            if (w > r[0]) w = r[0];
            const int h = r[2] > w + 1 ? w + 1 : r[2], cl = r[1] > w ? w : r[1];
            const uint32_t bytes = (4u << h) + (4u << cl);
            if (bytes > a.tableStride && bytes <= (12u << 17)) a.tableStride = bytes;
        }
    a.e1Lanes = g_hasCD ? ZE_E1_LANES_DICT : ZE_E1_LANES;
    a.slotSrcMax = g_hasCD ? g_dictSlotMax : 0u;
    a.laneTables = (uint8_t*)malloc((size_t)nBlocks * a.e1Lanes * a.tableStride);
    a.meta = (ZeMeta*)calloc(chunk, sizeof(ZeMeta));
    a.arena = (uint8_t*)malloc((size_t)chunk * ZE_ARENA_STRIDE); a.arenaStride = (uint32_t)ZE_ARENA_STRIDE; a.arenaLit = ZE_ARENA_LIT;
    uint32_t bigCount = 0; a.bigList = (uint32_t*)calloc(n ? n : 1, 4); a.bigCount = &bigCount;
    attach_cdict(a);
    memset(&g_elds, 0xA5, sizeof g_elds);
    bool anyDfast = false; for (int t = 2; t < 4; t++) anyDfast |= a.rows.r[t][6] == 2;
    const bool flatDict = g_hasCD && a.cdict && a.cdict->strat == 2;      // mirrors zhip_compress_batch_device
    const bool flat = (anyDfast && !g_hasCD) || flatDict;
    uint32_t e1Count = 0; a.e1List = (uint32_t*)calloc(chunk, 4); a.e1Count = &e1Count; a.useE1List = flat ? 1u : 0u;
    // launch numbers in the dictionary search's cells (ZhipEncodeArgs.tabEpoch; mirrors zhip_compress_batch_device): the tables persist from call to call, zeroed when they are
    // (re)made, and every launch on them takes the next number -- earlier launches' cells must read as empty
    const bool mbcWanted = flat && !flatDict && g_mbCompress && maxSrc > ZF_BLOCK_MAX && maxSrc < (1ull << ZE_MB_POS_BITS) - 8;
    const bool epochs = flat && !mbcWanted && g_dictEpochs;
    if (epochs) {
        const size_t need = (size_t)chunk * a.tableStride;
        uint32_t es = 26;                                      // (the dictionary-less search: six fixed bits, ze_dfast_flat_np)
        if (flatDict) { const uint64_t span = 2ull + a.cdict->contentSize + (a.slotSrcMax ? a.slotSrcMax : (uint32_t)ZE_DICT_ATTACH_MAX) + 64; es = 1; while ((1ull << es) < span) es++; }
        if (need > g_epochCap || es != g_epochShift || g_epoch + 8 >= (1u << (32 - es))) { free(g_epochTables); g_epochTables = (uint8_t*)calloc(need, 1); g_epochCap = need; g_epochShift = es; g_epoch = 0; }
        a.flatTables = g_epochTables; a.tabEpochShift = es;
    } else
    a.flatTables = flat ? (uint8_t*)malloc((size_t)chunk * a.tableStride) : nullptr;
    static uint8_t idlePad[64]; a.idle = idlePad;                  // (the product points it at the context's counter block)
    // sources of several blocks in the flat kernel (mirrors zhip_compress_batch_device: the size hint is the batch's largest source)
    const bool mbc = flat && !flatDict && g_mbCompress && maxSrc > ZF_BLOCK_MAX && maxSrc < (1ull << ZE_MB_POS_BITS) - 8;
    if (getenv("ZHIP_EMU_DEBUG")) fprintf(stderr, "[emu] mbc %d flat %d maxSrc %llu\n", (int)mbc, (int)flat, (unsigned long long)maxSrc);
    if (mbc) {
        const uint64_t hint = g_mbHint ? g_mbHint : maxSrc;
        a.mbMaxBlocks = g_mbCompress > 1 ? g_mbCompress : (uint32_t)(2 * ((hint + ZF_BLOCK_MAX - 1) / ZF_BLOCK_MAX) + 2); a.mbSeqCap = (uint32_t)(hint / 4 + a.mbMaxBlocks + 64);
        a.mbBlocks = (ZeMbBlock*)malloc((size_t)chunk * a.mbMaxBlocks * sizeof(ZeMbBlock)); a.mbCount = (uint32_t*)malloc((size_t)chunk * 4); a.mbSeqs = (uint64_t*)malloc((size_t)chunk * a.mbSeqCap * 8);
        memset(a.mbBlocks, 0xA5, (size_t)chunk * a.mbMaxBlocks * sizeof(ZeMbBlock)); memset(a.mbCount, 0xA5, (size_t)chunk * 4);
    }
    for (uint32_t first = 0; first < n; first += chunk) {
        a.first = first; a.count = n - first < chunk ? n - first : chunk;
        counters[0] = counters[1] = 0; e1Count = 0;
        if (flat) {
            if (epochs) a.tabEpoch = ++g_epoch;
            else memset(a.flatTables, flatDict ? 0xA5 : 0, (size_t)a.count * a.tableStride);      // (dictionary batches: the kernel's waves zero what they use)
            if (mbc) { memset(&g_elds, 0xA5, sizeof g_elds); zhemu::run_grid(a.count < 3 ? a.count : 3, split_lane, &a); if (getenv("ZHIP_EMU_DEBUG")) fprintf(stderr, "[emu] split: count[0] = %u stride %u\n", a.mbCount[0], a.tableStride); }
            if (a.count <= g_e1LdsMax && !flatDict && !mbc) { memset(&g_srclds, 0xA5, sizeof g_srclds); zhemu::run_grid(a.count, e1l_lane, &a); }
            else zhemu::run_grid((a.count + ZE_FLAT_LANES - 1) / ZE_FLAT_LANES, e1f_lane, &a);
            if (mbc) { a.mbLanes = 16; a.mbProbes = g_probes; zhemu::run_grid((a.count + a.mbLanes - 1) / a.mbLanes, e1fmb_lane, &a); }
        }
        zhemu::run_grid(nBlocks, e1_lane, &a);
        a.xxLater = g_xxLater && a.checksumFlag ? 1u : 0u;
        zhemu::run_grid(nBlocks, e2_lane, &a);
        if (a.xxLater) zhemu::run_grid(nBlocks, ex_lane, &a);
        a.xxLater = 0;
        if (mbc && bigCount) {                   // the chunk's sources of several blocks: the generic kernel right away (it reads the chunk's arenas)
            ZhipEncodeArgs b = a; uint32_t bc = 0;
            b.workspace = (uint8_t*)malloc((size_t)nBlocks * ZHIP_ENC_STRIDE); b.counter = &bc; b.frameList = a.bigList; b.listCount = &bigCount;
            EncLaunch l = { &b };
            zhemu::run_grid(nBlocks, enc_lane, &l);
            free(b.workspace);
            bigCount = 0;
        }
    }
    free(a.e1List); if (!epochs) free(a.flatTables); free(a.mbBlocks); free(a.mbCount); free(a.mbSeqs);
    if (bigCount) {                              // inputs above one block: generic kernel over the list (mirrors zhip_compress_batch_device)
        ZhipEncodeArgs b = a; uint32_t bc = 0;
        b.workspace = (uint8_t*)malloc((size_t)nBlocks * ZHIP_ENC_STRIDE); b.counter = &bc; b.frameList = a.bigList; b.listCount = &bigCount;
        EncLaunch l = { &b };
        zhemu::run_grid(nBlocks, enc_lane, &l);
        free(b.workspace);
    }
    free(a.workspace); free(a.laneTables); free(a.meta); free(a.arena); free(a.bigList);
    return 0;
}
extern "C" long emu_stat(int i) { return i >= 0 && i < 16 ? zd_stat[i] : -1; }     // [15]: frames searched by the flat match kernel
// exhaustive check of the computed LL / ML codes and extra-bit counts against the format's base tables (RFC 8878 3.1.1.3.2.1.1)
extern "C" int emu_check_code_formulas(void)
{
    int bad = 0;
    for (uint32_t v = 0; v < 131072; v++) {          // a sequence of a 128 KiB block has at most 131 069 literals
        uint32_t c = 35; while (ze_llBase[c] > v) c--;
        if (ze_ll_code(v) != c || ze_ll_bits(c) != ze_llBits[c]) bad++;
    }
    for (uint32_t ml = 3; ml <= 131074; ml++) {
        uint32_t c = 52; while (ze_mlBase[c] > ml) c--;
        if (ze_ml_code(ml) != c || ze_ml_bits(c) != ze_mlBits[c]) bad++;
    }
    return bad;
}

// ---- the entropy kernel's wave-parallel table builders on their own (checked against the oracle's serial restatement on arbitrary histograms)
static ZeLDS g_tbL;
struct TbFse { uint32_t count[64]; uint32_t total, maxSym, lg; int useLow; int16_t norm[64]; uint8_t out[128]; uint32_t h; ZeCTab tab; int rc; };
static void tbfse_lane(void* p)
{
    TbFse* d = (TbFse*)p;
    const uint32_t lane = zh_lane();
    ZeLDS& L = g_tbL;
    L.cnt[0][lane] = d->count[lane];
    zh_sync();
    int16_t* norm = ze_norm_area(L);
    const int rc = ze_fse_normalize_wave(norm, d->lg, L.cnt[0], d->total, d->maxSym, d->useLow != 0);
    ze_fence(); zh_sync();
    if (lane == 0) d->rc = rc;
    if (rc < 0) return;
    d->norm[lane] = lane <= d->maxSym ? norm[lane] : 0;
    const uint32_t h = ze_fse_write_ncount_wave(d->out, norm, d->maxSym, d->lg, (uint32_t*)((uint8_t*)L.node + 2048));
    if (lane == 0) d->h = h;
    ze_fse_build_ctab_wave(L.tab[0], norm, d->maxSym, d->lg, ze_cell_sym(L), (uint16_t*)L.stack, ze_fill_area(L));
    zh_sync();
    if (lane == 0) d->tab = L.tab[0];
}
extern "C" int emu_fse_tables(const uint32_t* count, uint32_t maxSym, uint32_t total, uint32_t lg, int useLow,
                              int16_t* normOut, uint8_t* ncountOut, uint32_t* ncountSize, uint16_t* cellOf, uint16_t* next)
{
    static TbFse d; memset(&d, 0, sizeof d);
    for (uint32_t s = 0; s <= maxSym; s++) d.count[s] = count[s];
    d.total = total; d.maxSym = maxSym; d.lg = lg; d.useLow = useLow;
    memset(&g_tbL, 0xA5, sizeof g_tbL);
    zhemu::run_grid(1, tbfse_lane, &d);
    if (d.rc < 0) return -1;
    memcpy(normOut, d.norm, sizeof d.norm); memcpy(ncountOut, d.out, d.h); *ncountSize = d.h;
    for (uint32_t s = 0; s <= maxSym + 1; s++) cellOf[s] = d.tab.cellOf[s];
    for (uint32_t u = 0; u < (1u << lg); u++) next[u] = d.tab.next[u];
    return 0;
}
struct TbHuf { uint32_t hist[256]; uint32_t maxSym, maxBits; uint8_t bits[256]; uint16_t code[256]; uint32_t lg; };
static void tbhuf_lane(void* p)
{
    TbHuf* d = (TbHuf*)p; ZeLDS& L = g_tbL; const uint32_t lane = zh_lane();
    for (uint32_t i = lane; i < 256; i += 64) L.hist[i] = d->hist[i];
    zh_sync();
    const uint32_t lg = ze_huf_build(L, d->maxSym, d->maxBits);
    zh_sync();
    if (lane == 0) { d->lg = lg; memcpy(d->bits, L.hufBits, 256); memcpy(d->code, L.hufCode, 512); }
}
extern "C" uint32_t emu_huf_build(const uint32_t* hist, uint32_t maxSym, uint32_t maxBits, uint8_t* bits, uint16_t* code)
{
    static TbHuf d; memset(&d, 0, sizeof d); memcpy(d.hist, hist, 1024); d.maxSym = maxSym; d.maxBits = maxBits;
    memset(&g_tbL, 0xA5, sizeof g_tbL);
    zhemu::run_grid(1, tbhuf_lane, &d);
    memcpy(bits, d.bits, 256); memcpy(code, d.code, 512);
    return d.lg;
}
