// zhip_cparams.hpp -- HOST side: compression level -> parameter rows (what ZSTD_getCParams / ZSTD_getCParamsFromCCtxParams compute).
//
// The level table is data of the format's reference implementation (ZSTD_defaultCParameters, zstd/zstd.c:30650-30755: 4 source-size
// classes x levels 0..22, columns windowLog chainLog hashLog searchLog minMatch targetLength strategy); frames are only bit-identical
// to libzstd's if the same numbers go in, so they are restated here as numbers (strategy: 1 fast 2 dfast 3 greedy 4 lazy 5 lazy2
// 6 btlazy2 7 btopt 8 btultra 9 btultra2). The kernels implement strategies 1 and 2; rows that resolve to anything else are refused
// per frame (ZE_PARAM_UNSUPPORTED), never approximated.
#pragma once
#include <stdint.h>
#include "../../include/zstd_hip.h"
#include "zhip_format.hpp"

static const int16_t zh_levelTable[4][23][7] = {
  {{19,12,13,1,6,1,1}, {19,13,14,1,7,0,1}, {20,15,16,1,6,0,1}, {21,16,17,1,5,0,2}, {21,18,18,1,5,0,2}, {21,18,19,3,5,2,3}, {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5}, {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5}, {22,22,23,6,5,32,5}, {22,22,22,4,5,32,6}, {22,22,23,5,5,32,6}, {22,23,23,6,5,32,6}, {22,22,22,5,5,48,7}, {23,23,22,5,4,64,7}, {23,23,22,6,3,64,8}, {23,24,22,7,3,256,9}, {25,25,23,7,3,256,9}, {26,26,24,7,3,512,9}, {27,27,25,9,3,999,9}},
  {{18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3}, {18,17,18,5,5,2,3}, {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5}, {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6}, {18,19,19,7,4,12,6}, {18,18,19,4,4,16,7}, {18,18,19,4,3,32,7}, {18,18,19,6,3,128,7}, {18,19,19,6,3,128,8}, {18,19,19,8,3,256,8}, {18,19,19,6,3,128,9}, {18,19,19,8,3,256,9}, {18,19,19,10,3,512,9}, {18,19,19,12,3,512,9}, {18,19,19,13,3,999,9}},
  {{17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,15,16,2,5,0,2}, {17,17,17,2,4,0,2}, {17,16,17,3,4,2,3}, {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5}, {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6}, {17,18,17,7,4,12,6}, {17,18,17,3,4,12,7}, {17,18,17,4,3,32,7}, {17,18,17,6,3,256,7}, {17,18,17,6,3,128,8}, {17,18,17,8,3,256,8}, {17,18,17,10,3,512,8}, {17,18,17,5,3,256,9}, {17,18,17,7,3,512,9}, {17,18,17,9,3,512,9}, {17,18,17,11,3,999,9}},
  {{14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,14,15,2,4,0,2}, {14,14,14,4,4,2,3}, {14,14,14,3,4,4,4}, {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6}, {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7}, {14,15,14,4,3,24,7}, {14,15,14,5,3,32,8}, {14,15,15,6,3,64,8}, {14,15,15,7,3,256,8}, {14,15,15,5,3,48,9}, {14,15,15,6,3,128,9}, {14,15,15,7,3,256,9}, {14,15,15,8,3,256,9}, {14,15,15,8,3,512,9}, {14,15,15,9,3,512,9}, {14,15,15,10,3,999,9}},
};

static inline uint32_t zh_hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// row of one size class for `level` with the caller's explicit fields laid over it (ZSTD_getCParams_internal zstd.c:30848 picks the row,
// ZSTD_overrideCParams :24578 applies the non-zero explicit fields); the per-source adjustment happens on the device (ze_get_cparams)
static inline void zh_resolve_row(int32_t out[7], int tableID, int level, const zhip_compression_parameters* ov)
{
    if (level == 0) level = 3;                                    // ZSTD_CLEVEL_DEFAULT
    const int row = level < 0 ? 0 : level > 22 ? 22 : level;
    for (int k = 0; k < 7; k++) out[k] = zh_levelTable[tableID][row][k];
    if (level < 0) { const int64_t t = -(int64_t)level; out[5] = (int32_t)(t > (1 << 17) ? (1 << 17) : t); }   // acceleration: targetLength = -level (zstd.c:30870)
    if (ov) {
        if (ov->windowLog) out[0] = (int32_t)ov->windowLog;
        if (ov->chainLog) out[1] = (int32_t)ov->chainLog;
        if (ov->hashLog) out[2] = (int32_t)ov->hashLog;
        if (ov->searchLog) out[3] = (int32_t)ov->searchLog;
        if (ov->minMatch) out[4] = (int32_t)ov->minMatch;
        if (ov->targetLength) out[5] = (int32_t)ov->targetLength;
        if (ov->strategy) out[6] = ov->strategy;
    }
}
static inline void zh_resolve_rows(ZeRows* rows, int level, const zhip_compression_parameters* ov)
{
    for (int t = 0; t < 4; t++) zh_resolve_row(rows->r[t], t, level, ov);
}
// bounds of ZSTD_checkCParams (zstd.c:24344; 64-bit limits of zstd.h): false = "Parameter is out of bound"
static inline bool zh_check_cparams(const int32_t r[7])
{
    return r[0] >= 10 && r[0] <= 31 && r[1] >= 6 && r[1] <= 30 && r[2] >= 6 && r[2] <= 30 && r[3] >= 1 && r[3] <= 30 &&
           r[4] >= 3 && r[4] <= 7 && r[5] >= 0 && r[5] <= (1 << 17) && r[6] >= 1 && r[6] <= 9;
}

// ZSTD_getCParams (zstd.c:30863 -> ZSTD_getCParams_internal :30848, mode ZSTD_cpm_unknown) followed by ZSTD_adjustCParams_internal
// (:24427): what ZstdCompressionParameters.from_level shows (c-ext/compressionparams.c:231-345). srcSizeHint 0 = unknown.
static inline void zh_get_cparams(int level, uint64_t srcSizeHint, size_t dictSize, zhip_compression_parameters* out)
{
    const bool unknown = srcSizeHint == 0;
    // (ZSTD_getCParamRowSize, zstd.c:30820: with the source size unknown libzstd adds ZSTD_CONTENTSIZE_UNKNOWN + dictSize + 500 in 64 bits,
    // which WRAPS to dictSize + 499 -- the row changes at dictSize 15 885 / 130 573 / 261 645, not one byte later)
    const uint64_t added = unknown && dictSize > 0 ? 499 : 0;
    const uint64_t rSize = unknown && dictSize == 0 ? ~0ull : srcSizeHint + dictSize + added;
    const int tableID = (rSize <= 256u * 1024) + (rSize <= 128u * 1024) + (rSize <= 16u * 1024);
    int32_t r[7];
    zh_resolve_row(r, tableID, level, nullptr);
    const uint64_t maxResize = 1ull << 30;
    if (!unknown && srcSizeHint <= maxResize && dictSize <= maxResize) {
        const uint32_t t = (uint32_t)(srcSizeHint + dictSize);
        const int32_t srcLog = t < 64 ? 6 : (int32_t)zh_hb32(t - 1) + 1;
        if (r[0] > srcLog) r[0] = srcLog;
    }
    if (!unknown) {
        int32_t dawl = r[0];
        if (dictSize) {
            const uint64_t win = 1ull << r[0], both = dictSize + win;
            if (win >= dictSize + srcSizeHint) dawl = r[0];
            else if (both >= (1ull << 31)) dawl = 31;
            else dawl = (int32_t)zh_hb32((uint32_t)both - 1) + 1;
        }
        const int32_t cycleLog = r[1] - (r[6] >= 6 ? 1 : 0);
        if (r[2] > dawl + 1) r[2] = dawl + 1;
        if (cycleLog > dawl) r[1] -= cycleLog - dawl;
    }
    if (r[0] < 10) r[0] = 10;
    // (the row-match-finder clamp of hashLog, zstd.c:24546-24556, cannot bind: hashLog <= 30 < 24 + 4 only matters above 28)
    if (r[6] >= 3 && r[6] <= 5) { const int32_t rowLog = r[3] < 4 ? 4 : r[3] > 6 ? 6 : r[3]; if (r[2] > 24 + rowLog) r[2] = 24 + rowLog; }
    out->windowLog = (uint32_t)r[0]; out->chainLog = (uint32_t)r[1]; out->hashLog = (uint32_t)r[2]; out->searchLog = (uint32_t)r[3];
    out->minMatch = (uint32_t)r[4]; out->targetLength = (uint32_t)r[5]; out->strategy = r[6];
}
