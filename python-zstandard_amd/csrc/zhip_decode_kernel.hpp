// zhip_decode_kernel.hpp -- batch zstd frame decoder, one wavefront (64 lanes) per frame.
//
// Replaces the arithmetic under decompress_worker() (c-ext/decompressor.c:944-1181), i.e. libzstd's
// ZSTD_decompressFrame (zstd.c:44174) and everything below it, for thousands of
// independent frames at once. Written from the format (RFC 8878); structure is CDNA-first:
//
//   * persistent workgroups of ONE wave pull frame indices from an atomic counter (dynamic balance: frame cost
//     varies 100x between a raw block and a text block);
//   * FSE decode tables (LL/ML/OF, <=5 KiB) and the Huffman table (<=8 KiB) live in LDS; LL/ML baselines too;
//   * literals: the 4 Huffman streams are decoded by 4 lanes into a per-wave scratch buffer that stays in L2;
//   * sequences: the backward bitstream is staged into LDS 512 bytes at a time with coalesced 8-byte loads;
//     lane 0 runs the three interleaved tANS states (the only inherently serial chain) and emits up to 64
//     (litLen, matchLen, offset) triples into LDS;
//   * execution is wave-parallel: a prefix scan turns the 64 triples into output positions, every lane copies its
//     own literals, and matches are resolved in dependency rounds (a match is copied as soon as its whole source
//     range lies below the first still-pending match); long copies and RLE-style overlaps go wave-wide.
//
// The same source compiles for the host under ZHIP_EMU (tests/emu) to debug logic without a GPU.
#pragma once
#include "zhip_device.hpp"
#include "zhip_format.hpp"
#include "zhip_xxh64.hpp"

#if defined(ZHIP_EMU) && defined(ZD_TRACE)
#include <stdio.h>
extern "C" { extern long zd_trace_pos; extern long zd_cur_frame; extern long zd_stat[16]; }
#define ZD_STAT(i, v) do { if (zh_lane() == 0) zd_stat[i] += (v); } while (0)
#define ZD_TRR(lo, n, what, a1, a2) do { if (zh_lane() == 0 && zd_trace_pos + 40 >= (long)(lo) && zd_trace_pos < (long)(lo) + (long)(n) + 40) fprintf(stderr, "[trace] f%ld range [%ld,+%ld) %s %ld %ld\n", zd_cur_frame, (long)(lo), (long)(n), what, (long)(a1), (long)(a2)); } while (0)
#define ZD_TR(pos, what, a1, a2, a3) do { if ((long)(pos) == zd_trace_pos) fprintf(stderr, "[trace] f%ld pos=%ld %s lane=%u %ld %ld %ld\n", zd_cur_frame, (long)(pos), what, zh_lane(), (long)(a1), (long)(a2), (long)(a3)); } while (0)
#else
#define ZD_TR(pos, what, a1, a2, a3) do { } while (0)
#define ZD_TRR(lo, n, what, a1, a2) do { } while (0)
#define ZD_STAT(i, v) do { } while (0)
#endif
#define ZD_STAGE_BYTES 512
#define ZD_ASM_BYTES 4096       // output bytes assembled in LDS per batch before the coalesced flush
#define ZD_COOP_LEN 32          // copies longer than this are done by the whole wave

ZH_CONST uint32_t zc_llBase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,
                                   1024,2048,4096,8192,16384,32768,65536};
ZH_CONST uint8_t zc_llBits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
ZH_CONST uint32_t zc_mlBase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,
                                   33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
ZH_CONST uint8_t zc_mlBits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                  1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
ZH_CONST int16_t zc_llDef[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
ZH_CONST int16_t zc_mlDef[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                 1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
ZH_CONST int16_t zc_ofDef[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

// FSE decode cell, one dword: base[0:10) nbBits[10:14) extraBits[14:19) symbol[19:25)
#define ZD_CELL(base, nb, extra, sym) ((uint32_t)(base) | ((uint32_t)(nb) << 10) | ((uint32_t)(extra) << 14) | ((uint32_t)(sym) << 19))
#define ZD_FSE_LL 0
#define ZD_FSE_ML 512
#define ZD_FSE_OF 1024
enum { ZD_KIND_LL = 0, ZD_KIND_OF = 1, ZD_KIND_ML = 2, ZD_KIND_W = 3 };

struct ZdLDS {
    uint32_t fse[1280];
    union alignas(16) {
        uint16_t huf[4096];
        struct { uint32_t stage[ZD_STAGE_BYTES / 4 + 4]; uint32_t ll[64], ml[64], of[64]; } q;
        struct { uint32_t wfse[64]; uint8_t pad[2048 - 256]; uint8_t symAt[512]; } b;
        struct { uint8_t pad[2048]; uint8_t asmb[ZD_ASM_BYTES + 64]; } a;   // batch output assembly buffer
    } u;
    uint32_t llBase[36];
    uint32_t mlBase[53];
    uint8_t llBits[36];
    uint8_t mlBits[56];
    uint8_t weights[256];
    int16_t norm[64];
    uint16_t run[64];
    uint16_t ends[64];
    uint32_t misc[8];
};

// per-frame decoder state (wave-uniform values, kept in registers)
struct ZdState {
    uint32_t rep0, rep1, rep2;
    uint32_t hufCount;            // weights incl. implied last; 0 = no Huffman table yet
    uint32_t llLog, ofLog, mlLog; // 0xFF = table not valid
    const uint8_t* litPtr;        // literal source for this block (src or scratch)
    uint32_t litSize;
    uint32_t litRLE;              // 1: all literals == rleByte
    uint32_t rleByte;
};

// ------------------------------------------------------------------------------------------ phase timers (tuning aid)
enum { ZP_HEADER = 0, ZP_HUFTAB, ZP_HUFDEC, ZP_SEQTAB, ZP_STAGE, ZP_SEQDEC, ZP_EXEC1, ZP_EXEC2, ZP_FLUSH, ZP_RAW, ZP_N };
struct ZdProf { bool on; uint64_t t0; uint64_t acc[ZP_N]; };
#ifndef ZHIP_EMU
ZH_DEV uint64_t zd_clock() { return __builtin_readcyclecounter(); }
ZH_DEV uint64_t zd_wall_clock() { return (uint64_t)wall_clock64(); }          // constant 100 MHz: comparable across waves and CUs (the pick study's per-wave durations)
#else
ZH_DEV uint64_t zd_clock() { return 0; }
ZH_DEV uint64_t zd_wall_clock() { return 0; }
#endif
#define ZD_T(P, i) do { if ((P).on) { uint64_t t1_ = zd_clock(); (P).acc[i] += t1_ - (P).t0; (P).t0 = t1_; } } while (0)

// ------------------------------------------------------------------------------------------ small helpers
ZH_DEV uint64_t zd_ld64_bounded(const uint8_t* p, const uint8_t* end)
{
    if (p + 8 <= end) return zh_ld64(p);
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) if (p + i < end) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// len bytes by the whole wave (ranges must not overlap): 16 bytes per lane per step with four steps' loads in flight before the
// first store. The obvious `dst[j] = src[j]` loop is one memory round trip per 64 bytes, because the compiler has to assume the
// store feeds the next load: 2 048 round trips for a raw 128 KiB block.
ZH_DEV void zd_copy_wave(uint8_t* dst, const uint8_t* src, uint32_t len)
{
    const uint32_t lane = zh_lane();
    const uint32_t ng = len >> 4;
    for (uint32_t g = lane; g < ng; g += 256) {
        zh_v16 v0 = zh_ld128(src + 16 * (size_t)g), v1 = v0, v2 = v0, v3 = v0;
        if (g + 64 < ng) v1 = zh_ld128(src + 16 * (size_t)(g + 64));
        if (g + 128 < ng) v2 = zh_ld128(src + 16 * (size_t)(g + 128));
        if (g + 192 < ng) v3 = zh_ld128(src + 16 * (size_t)(g + 192));
        *(zh_v16*)(dst + 16 * (size_t)g) = v0;
        if (g + 64 < ng) *(zh_v16*)(dst + 16 * (size_t)(g + 64)) = v1;
        if (g + 128 < ng) *(zh_v16*)(dst + 16 * (size_t)(g + 128)) = v2;
        if (g + 192 < ng) *(zh_v16*)(dst + 16 * (size_t)(g + 192)) = v3;
    }
    const uint32_t done = ng << 4;
    if (lane < len - done) dst[done + lane] = src[done + lane];
}
ZH_DEV void zd_fill_wave(uint8_t* dst, uint32_t byte, uint32_t len)
{
    const uint32_t lane = zh_lane();
    const uint32_t ng = len >> 4;
    zh_v16 v; v.lo = 0x0101010101010101ull * (byte & 255); v.hi = v.lo;
    for (uint32_t g = lane; g < ng; g += 64) *(zh_v16*)(dst + 16 * (size_t)g) = v;
    const uint32_t done = ng << 4;
    if (lane < len - done) dst[done + lane] = (uint8_t)byte;
}
// len bytes from global memory into the LDS assembly buffer (any alignment, so bytes), eight per lane in flight per step
ZH_DEV void zd_stage_wave(uint8_t* lds, const uint8_t* src, uint32_t len)
{
    for (uint32_t j = zh_lane(); j < len; j += 512) {
        uint8_t b[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) b[k] = j + 64 * k < len ? src[j + 64 * k] : (uint8_t)0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) if (j + 64 * k < len) lds[j + 64 * k] = b[k];
    }
}

#ifndef ZHIP_EMU
ZH_DEV void zd_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
#else
ZH_DEV void zd_fence() { zhemu::collective_wait(); }   // emulated lanes are not lockstep: a fence orders them like the hardware's in-order wave does
#endif

// ------------------------------------------------------------------------------------------ FSE tables
// forward bit reader used only by lane 0 for table descriptions (a few dozen bytes)
// (round 4: the reader keeps the 8 bytes around its cursor in registers and loads again only when a field would run past them -- every ~5
// fields instead of one or two loads per field: the parse is a serial chain of lane 0 and K1's time is the latency of that chain)
struct ZdFwd { const uint8_t* p; const uint8_t* end; uint32_t bitpos; uint64_t w; uint32_t wbit; };      // w = the 8 bytes at bit offset wbit (a multiple of 8)
ZH_DEV void zd_fwd_init(ZdFwd& f, const uint8_t* p, const uint8_t* end) { f.p = p; f.end = end; f.bitpos = 0; f.wbit = 0; f.w = zd_ld64_bounded(p, end); }
ZH_DEV uint32_t zd_fwd_peek(ZdFwd& f, uint32_t n)                     // n <= 16
{
    uint32_t off = f.bitpos - f.wbit;
    if (off + n > 64) { f.wbit = f.bitpos & ~7u; f.w = zd_ld64_bounded(f.p + (f.wbit >> 3), f.end); off = f.bitpos - f.wbit; }
    return (uint32_t)(f.w >> off) & ((1u << n) - 1);
}

// One lane. Parses an FSE distribution (RFC 8878 4.1.1) into norm[0 .. 64). Returns bytes used or -err.
// *pMax in: alphabet limit, out: last symbol present. *pLog out.
// (K1 calls it on lane 0 with the wave's L.norm; K0 -- zp_pre_body -- on every lane with the lane's own record)
ZH_DEVFN int zd_read_ncount_to(int16_t* norm, const uint8_t* src, const uint8_t* end, uint32_t* pMax, uint32_t* pLog)
{
    if (src >= end) return -ZE_SRC_SIZE_WRONG;
    ZdFwd f; zd_fwd_init(f, src, end);
    uint32_t srcBits = (uint32_t)(end - src) * 8;
    int al = (int)zd_fwd_peek(f, 4) + 5; f.bitpos += 4;
    if (al > 15) return -ZE_TABLELOG_TOO_LARGE;
    *pLog = (uint32_t)al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbBits = al + 1;
    uint32_t sym = 0, maxS = *pMax;
    int prev0 = 0;
    while (remaining > 1 && sym <= maxS) {
        if (prev0) {
            for (;;) {
                uint32_t r = zd_fwd_peek(f, 2); f.bitpos += 2;
                for (uint32_t k = 0; k < r && sym <= maxS; k++) norm[sym++] = 0;
                if (r != 3) break;
                if (f.bitpos > srcBits) return -ZE_CORRUPTION;
            }
            if (sym > maxS) return -ZE_MAXSYMBOL_TOO_SMALL;
        }
        int max = (2 * threshold - 1) - remaining;
        int count;
        int low = (int)zd_fwd_peek(f, (uint32_t)nbBits - 1);
        if (low < max) { count = low; f.bitpos += (uint32_t)nbBits - 1; }
        else {
            count = (int)zd_fwd_peek(f, (uint32_t)nbBits);
            if (count >= threshold) count -= max;
            f.bitpos += (uint32_t)nbBits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        if (remaining < 1) return -ZE_CORRUPTION;
        norm[sym++] = (int16_t)count;
        prev0 = (count == 0);
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if (((f.bitpos + 7) >> 3) > (uint32_t)(end - src)) return -ZE_CORRUPTION;
    }
    if (remaining != 1) return -ZE_CORRUPTION;
    for (uint32_t s = sym; s < 64; s++) norm[s] = 0;
    *pMax = sym - 1;
    return (int)((f.bitpos + 7) >> 3);
}
ZH_DEV int zd_read_ncount(ZdLDS& L, const uint8_t* src, const uint8_t* end, uint32_t* pMax, uint32_t* pLog) { return zd_read_ncount_to(L.norm, src, end, pMax, pLog); }

// Wave-parallel construction of an FSE decoding table from L.norm[0..maxSym] (all lanes call).
// One lane per symbol for the bookkeeping, one lane per cell for spreading and state numbering.
ZH_DEVFN int zd_build_fse(ZdLDS& L, uint32_t* table, uint32_t maxSym, uint32_t log, int kind)
{
    const uint32_t lane = zh_lane();
    const uint32_t S = 1u << log, mask = S - 1, step = (S >> 1) + (S >> 3) + 3;
    const uint64_t lt = zh_lt_mask();
    zh_sync();                                   // L.norm written by lane 0 / other lanes
    int n = (lane <= maxSym) ? (int)L.norm[lane] : 0;
    bool low = (n == -1);
    uint64_t lowMask = zh_ballot(low);
    uint32_t nLow = (uint32_t)zh_popc64(lowMask);
    uint32_t high = S - 1 - nLow;
    if (low) L.u.b.symAt[S - 1 - (uint32_t)zh_popc64(lowMask & lt)] = (uint8_t)lane;
    uint32_t cnt = n > 0 ? (uint32_t)n : 0;
    uint32_t incl = zh_scan_add(cnt);
    uint32_t total = zh_shfl(incl, 63);
    if (total + nLow != S) return -ZE_CORRUPTION;
    L.ends[lane] = (uint16_t)incl;
    L.run[lane] = (uint16_t)(low ? 1 : cnt);
    zh_sync();
    // spread: the k-th visited position is (k*step)&mask; positions above `high` are skipped
    uint32_t jbase = 0;
    for (uint32_t c = 0; c < S; c += 64) {
        uint32_t k = c + lane;
        uint32_t p = (k * step) & mask;
        bool valid = (k < S) && (p <= high);
        uint64_t m = zh_ballot(valid);
        if (valid) {
            uint32_t j = jbase + (uint32_t)zh_popc64(m & lt);
            uint32_t pos = 0;                   // first symbol whose inclusive end exceeds j
            for (uint32_t stepb = 32; stepb; stepb >>= 1)
                if (L.ends[pos + stepb - 1] <= j) pos += stepb;
            L.u.b.symAt[p] = (uint8_t)pos;
        }
        jbase += (uint32_t)zh_popc64(m);
    }
    zh_sync();
    // number the cells of every symbol in table order: x = norm + rank, nbBits = log - highbit(x)
    for (uint32_t c = 0; c < S; c += 64) {
        uint32_t u = c + lane;
        bool act = u < S;
        uint32_t s = act ? L.u.b.symAt[u] : 0;
        uint64_t same = zh_ballot(act);
        for (int b = 0; b < 6; b++) {
            uint64_t bm = zh_ballot(((s >> b) & 1) != 0);
            same &= ((s >> b) & 1) ? bm : ~bm;
        }
        uint32_t x = 0;
        if (act) x = (uint32_t)L.run[s] + (uint32_t)zh_popc64(same & lt);
        zh_sync();
        if (act) {
            if ((same >> lane) >> 1 == 0) L.run[s] = (uint16_t)(x + 1);   // highest lane of its group
            uint32_t nb = log - (uint32_t)zh_highbit32(x);
            uint32_t base = (x << nb) - S;
            uint32_t extra = kind == ZD_KIND_LL ? L.llBits[s] : kind == ZD_KIND_ML ? L.mlBits[s] : kind == ZD_KIND_OF ? s : 0;
            table[u] = ZD_CELL(base, nb, extra, s);
        }
        zh_sync();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ backward bit reader (global)
struct ZdBits { const uint8_t* start; const uint8_t* ptr; uint64_t c; uint32_t used; };

ZH_DEV bool zd_bits_init(ZdBits& b, const uint8_t* src, uint32_t size)
{
    b.start = src; b.ptr = src; b.c = 0; b.used = 64;
    if (size == 0) return false;
    uint32_t last = src[size - 1];
    if (last == 0) return false;
    uint32_t pad = 8 - (uint32_t)zh_highbit32(last);
    if (size >= 8) { b.ptr = src + size - 8; b.c = zh_ld64(b.ptr); b.used = pad; }
    else {
        uint64_t c = 0;
        for (uint32_t i = 0; i < size; i++) c |= (uint64_t)src[i] << (8 * i);
        b.c = c; b.used = pad + (8 - size) * 8;
    }
    return true;
}
ZH_DEV void zd_bits_reload(ZdBits& b)
{
    uint32_t nb = b.used >> 3;
    uint32_t room = (uint32_t)(b.ptr - b.start);
    if (nb > room) nb = room;
    if (nb) { b.ptr -= nb; b.used -= nb * 8; b.c = zh_ld64(b.ptr); }
}
ZH_DEV uint32_t zd_bits_peek(const ZdBits& b, uint32_t n)   // 1 <= n <= 32
{
    return (uint32_t)((b.c << (b.used & 63)) >> (64 - n));
}
ZH_DEV bool zd_bits_done(const ZdBits& b) { return b.ptr == b.start && b.used == 64; }

// ------------------------------------------------------------------------------------------ prefetching backward reader
// Same stream convention as ZdBits, but the NEXT 8 bytes below the window are always already in flight (`d`), so the
// per-symbol / per-sequence critical path never waits for global memory: renormalisation is a funnel shift of (c, d)
// followed by issuing the load that will only be needed one renormalisation later.
struct ZdPBits { const uint8_t* start; const uint8_t* end; const uint8_t* ptr; uint64_t c, d; uint32_t used; };
ZH_DEV uint64_t zd_pb_below(const ZdPBits& b)           // bytes at [ptr-8, ptr), zeros below the stream start
{
    const uint32_t k = (uint32_t)(b.ptr - b.start);
    if (k >= 8) return zh_ld64(b.ptr - 8);
    if (k == 0) return 0;
    return zd_ld64_bounded(b.start, b.end) << (8 * (8 - k));
}
ZH_DEV bool zd_pb_init(ZdPBits& b, const uint8_t* src, uint32_t size)
{
    b.start = src; b.end = src + size; b.ptr = src; b.c = 0; b.d = 0; b.used = 64;
    if (size == 0) return false;
    const uint32_t last = src[size - 1];
    if (last == 0) return false;
    const uint32_t pad = 8 - (uint32_t)zh_highbit32(last);
    if (size >= 8) { b.ptr = src + size - 8; b.c = zh_ld64(b.ptr); b.used = pad; }
    else {
        uint64_t c = 0;
        for (uint32_t i = 0; i < size; i++) c |= (uint64_t)src[i] << (8 * i);
        b.c = c; b.used = pad + (8 - size) * 8;
    }
    b.d = zd_pb_below(b);
    return true;
}
ZH_DEV void zd_pb_norm(ZdPBits& b)
{
    uint32_t nb = b.used >> 3;
    const uint32_t room = (uint32_t)(b.ptr - b.start);
    if (nb > room) nb = room;
    if (nb > 8) nb = 8;
    if (nb) {
        b.c = nb == 8 ? b.d : ((b.c << (8 * nb)) | (b.d >> (64 - 8 * nb)));
        b.ptr -= nb; b.used -= nb * 8;
        b.d = zd_pb_below(b);
    }
}
ZH_DEV uint32_t zd_pb_peek(const ZdPBits& b, uint32_t n) { return (uint32_t)((b.c << (b.used & 63)) >> (64 - n)); }   // 1 <= n <= 32
ZH_DEV bool zd_pb_done(const ZdPBits& b) { return b.ptr == b.start && b.used == 64; }

// ------------------------------------------------------------------------------------------ Huffman
// Reads a tree description into L.weights (all lanes call). Returns bytes used or -err; *pCount = #weights incl. last.
// `pre` (K1 of the pipeline, round 6): K0's record of the frame -- when it holds the weights of the description at block offset `at`, they are copied and nothing is parsed
ZH_DEVFN int zd_read_huf_weights(ZdLDS& L, const uint8_t* src, uint32_t srcSize, uint32_t* pCount, uint32_t* pLog, const ZpPre* pre = nullptr, uint32_t at = 0)
{
    const uint32_t lane = zh_lane();
    if (srcSize < 1) return -ZE_CORRUPTION;
    uint32_t hb = src[0], n, used;
    if (hb >= 128) {
        n = hb - 127; used = 1 + (n + 1) / 2;
        if (used > srcSize) return -ZE_CORRUPTION;
        for (uint32_t i = lane; i < n; i += 64) {
            uint32_t v = src[1 + i / 2];
            L.weights[i] = (uint8_t)((i & 1) ? (v & 15) : (v >> 4));
        }
    } else {
        used = 1 + hb;
        if (used > srcSize || hb < 2) return -ZE_CORRUPTION;
        if (pre && pre->wCount && pre->wAt == at) {
            n = pre->wCount;
            zh_sync();
            ((uint32_t*)L.weights)[lane] = ((const uint32_t*)pre->weights)[lane];       // (all 256 bytes: what lies past the count is never read)
#ifdef ZHIP_EMU
            if (lane == 0) zd_stat[10]++;                                                // (test hook [10]: Huffman weights taken from K0's record)
#endif
        } else {
        // Round 4: the description (< 128 bytes) is copied to LDS by the whole wave and lane 0 parses THAT. Parsed in place, every field of the
        // distribution and every refill of the weights' bit reader below was a dependent global-memory round trip (~300 of them per frame with
        // the three sequence tables, ~500 cycles each): K1's time is frames per resident wave x the latency of ONE frame's chain (210 K cycles).
        // The staging area is the upper half of the union: free until zd_build_huf writes the table (wfse / symAt lie in its first 2.5 KiB).
        uint8_t* const stg = (uint8_t*)L.u.huf + 4096;
        zh_sync();
        zd_stage_wave(stg, src + 1, hb);
        zh_sync();
        if (lane == 0) {
            uint32_t maxS = 255 > 63 ? 63 : 255, tl = 0;     // weights alphabet is 0..12; 63 is plenty
            int r = zd_read_ncount(L, stg, stg + hb, &maxS, &tl);
            L.misc[0] = (uint32_t)r; L.misc[1] = maxS; L.misc[2] = tl;
        }
        zh_sync();
        int r = (int)L.misc[0]; uint32_t maxS = L.misc[1], tl = L.misc[2];
        if (r < 0 || tl > 6 || maxS > 12) return -ZE_CORRUPTION;
        if (zd_build_fse(L, L.u.b.wfse, maxS, tl, ZD_KIND_W) < 0) return -ZE_CORRUPTION;
        if (lane == 0) {
            ZdBits b; uint32_t cnt = 0; int bad = 0;
            if (!zd_bits_init(b, stg + r, hb - (uint32_t)r)) bad = 1;
            else {
                // two interleaved states; stop when the stream over-reads (RFC 8878 4.2.1.2)
                zd_bits_reload(b);
                uint32_t s1 = zd_bits_peek(b, tl); b.used += tl;
                uint32_t s2 = zd_bits_peek(b, tl); b.used += tl;
                if (tl == 0) { s1 = s2 = 0; }
                int64_t left;   // bits still unread (may go negative)
                for (;;) {
                    if (b.used >= 32) zd_bits_reload(b);            // (two weights take at most 12 bits: a refill every few trips, not a load in each)
                    left = (int64_t)(b.ptr - b.start) * 8 + 64 - (int64_t)b.used;
                    uint32_t e1 = L.u.b.wfse[s1], nb1 = (e1 >> 10) & 15;
                    if (cnt > 253) { bad = 1; break; }
                    L.weights[cnt++] = (uint8_t)(e1 >> 19);
                    uint32_t v1 = nb1 ? zd_bits_peek(b, nb1) : 0; b.used += nb1; left -= nb1;
                    s1 = (e1 & 1023) + v1;
                    uint32_t e2 = L.u.b.wfse[s2], nb2 = (e2 >> 10) & 15;
                    if (left < 0) { L.weights[cnt++] = (uint8_t)(e2 >> 19); break; }
                    if (cnt > 253) { bad = 1; break; }
                    L.weights[cnt++] = (uint8_t)(e2 >> 19);
                    uint32_t v2 = nb2 ? zd_bits_peek(b, nb2) : 0; b.used += nb2; left -= nb2;
                    s2 = (e2 & 1023) + v2;
                    if (left < 0) { L.weights[cnt++] = (uint8_t)(L.u.b.wfse[s1] >> 19); break; }
                }
            }
            L.misc[0] = bad ? 0xFFFFFFFFu : cnt;
        }
        zh_sync();
        n = L.misc[0];
        if (n == 0xFFFFFFFFu || n == 0) return -ZE_CORRUPTION;
        }
    }
    zh_sync();
    // implied last weight: the sum of 2^(w-1) must complete to a power of two
    uint32_t part = 0, ones = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        uint32_t w = L.weights[i];
        if (w > 12) part = 0x40000000u;
        else if (w) part += 1u << (w - 1);
        ones += (w == 1);
    }
    uint32_t total = zh_shfl(zh_scan_add(part), 63);
    uint32_t nOnes = zh_shfl(zh_scan_add(ones), 63);
    if (total == 0 || total >= 0x40000000u) return -ZE_CORRUPTION;
    uint32_t log = (uint32_t)zh_highbit32(total) + 1;
    if (log > 12) return -ZE_CORRUPTION;
    uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1)) return -ZE_CORRUPTION;
    uint32_t lastW = (uint32_t)zh_highbit32(rest) + 1;
    nOnes += (lastW == 1);
    if (nOnes < 2 || (nOnes & 1)) return -ZE_CORRUPTION;
    if (lane == 0) L.weights[n] = (uint8_t)lastW;
    zh_sync();
    *pCount = n + 1; *pLog = log;
    return (int)used;
}

// L.weights[0..count) -> L.u.huf[ 1<<log ] cells (sym | nbBits<<8). All lanes call. Returns log or -err.
ZH_DEVFN int zd_build_huf(ZdLDS& L, uint32_t count)
{
    const uint32_t lane = zh_lane();
    const uint64_t lt = zh_lt_mask();
    uint32_t wt[4], myStart[4];
    uint32_t part = 0;
    for (int k = 0; k < 4; k++) {
        uint32_t s = (uint32_t)k * 64 + lane;
        wt[k] = s < count ? L.weights[s] : 0;
        if (wt[k]) part += 1u << (wt[k] - 1);
        myStart[k] = 0;
    }
    uint32_t total = zh_shfl(zh_scan_add(part), 63);
    if (total == 0 || (total & (total - 1))) return -ZE_CORRUPTION;
    uint32_t log = (uint32_t)zh_highbit32(total);
    if (log > 12 || log == 0) return -ZE_CORRUPTION;
    zh_sync();                                           // previous users of the union are done
    uint32_t base = 0;
    for (uint32_t w = 1; w <= 12; w++) {
        uint32_t before = 0;
        for (int k = 0; k < 4; k++) {
            uint64_t m = zh_ballot(wt[k] == w);
            if (wt[k] == w) myStart[k] = base + ((before + (uint32_t)zh_popc64(m & lt)) << (w - 1));
            before += (uint32_t)zh_popc64(m);
        }
        base += before << (w - 1);
    }
    for (int k = 0; k < 4; k++) {
        if (!wt[k]) continue;
        uint32_t len = 1u << (wt[k] - 1);
        uint32_t val = ((uint32_t)k * 64 + lane) | ((log + 1 - wt[k]) << 8);
        if (len >= 4) {
            uint64_t v4 = (uint64_t)val * 0x0001000100010001ull;
            uint64_t* t = (uint64_t*)&L.u.huf[myStart[k]];
            for (uint32_t i = 0; i < len / 4; i++) t[i] = v4;
        } else {
            for (uint32_t i = 0; i < len; i++) L.u.huf[myStart[k] + i] = (uint16_t)val;
        }
    }
    zh_sync();
    return (int)log;
}

// one Huffman stream, one lane. Writes `count` symbols to out. Returns true when the stream was consumed exactly.
ZH_DEVFN bool zd_huf_stream(const uint16_t* huf, uint32_t log, const uint8_t* src, uint32_t srcSize, uint8_t* out, uint32_t count)
{
    ZdPBits b;
    if (!zd_pb_init(b, src, srcSize)) return false;
    uint32_t i = 0;
    zd_pb_norm(b);
    while (i + 4 <= count) {
        uint32_t e0 = huf[zd_pb_peek(b, log)]; b.used += e0 >> 8;
        uint32_t e1 = huf[zd_pb_peek(b, log)]; b.used += e1 >> 8;
        uint32_t e2 = huf[zd_pb_peek(b, log)]; b.used += e2 >> 8;
        uint32_t e3 = huf[zd_pb_peek(b, log)]; b.used += e3 >> 8;
        if (b.used > 64) return false;
        zh_st32(out + i, (e0 & 255) | ((e1 & 255) << 8) | ((e2 & 255) << 16) | ((e3 & 255) << 24));
        i += 4;
        zd_pb_norm(b);
    }
    while (i < count) {
        uint32_t e = huf[zd_pb_peek(b, log)]; b.used += e >> 8;
        if (b.used > 64) return false;
        out[i++] = (uint8_t)e;
    }
    zd_pb_norm(b);
    return zd_pb_done(b);
}

// Literals section (RFC 8878 3.1.1.3.1). All lanes call. Returns bytes consumed or -err.
// `defer` (pipeline K1 only): instead of decoding Huffman streams here, hand the table and the stream location to K1b.
struct ZdLitDefer { uint16_t* table; uint32_t maxLog; uint32_t taken, log, four, streamBytes; const uint8_t* streams;
                    const uint16_t* prevTable; uint32_t prevLog;      // prevTable: the "previous" table of a treeless block, ready-made (a dictionary's)
                    uint32_t shared, shareOK;                         // shareOK: the caller reads prevTable where it lies (no copy into `table`); shared: that happened
                    const ZpPre* pre; };                              // K0's record of the frame or null (zd_read_huf_weights)
ZH_DEVFN int zd_literals(ZdLDS& L, ZdState& st, const uint8_t* src, uint32_t srcSize, uint8_t* lit, uint32_t blockMax, ZdProf& P,
                         ZdLitDefer* defer = nullptr)
{
    const uint32_t lane = zh_lane();
    if (srcSize < 1) return -ZE_CORRUPTION;
    uint32_t b0 = src[0], type = b0 & 3, fmt = (b0 >> 2) & 3;
    uint32_t hdr, regen, csize = 0, four = 0;
    st.litRLE = 0;
    if (type < 2) {
        if (fmt == 1) { if (srcSize < 2) return -ZE_CORRUPTION; hdr = 2; regen = zh_ld16(src) >> 4; }
        else if (fmt == 3) { if (srcSize < 3) return -ZE_CORRUPTION; hdr = 3; regen = zh_ld24(src) >> 4; }
        else { hdr = 1; regen = b0 >> 3; }
        if (regen > blockMax) return -ZE_CORRUPTION;
        st.litSize = regen;
        if (type == 0) {
            if (hdr + regen > srcSize) return -ZE_CORRUPTION;
            st.litPtr = src + hdr;
            return (int)(hdr + regen);
        }
        if (hdr + 1 > srcSize) return -ZE_CORRUPTION;
        st.litRLE = 1; st.rleByte = src[hdr]; st.litPtr = lit;
        return (int)(hdr + 1);
    }
    if (srcSize < 5) return -ZE_CORRUPTION;
    uint32_t v = zh_ld32(src);
    if (fmt < 2) { hdr = 3; four = fmt; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
    else if (fmt == 2) { hdr = 4; four = 1; regen = (v >> 4) & 0x3FFF; csize = v >> 18; }
    else { hdr = 5; four = 1; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) + ((uint32_t)src[4] << 10); }
    if (regen > blockMax || regen > ZF_BLOCK_MAX) return -ZE_CORRUPTION;
    if (!four && regen == 0) return -ZE_CORRUPTION;
    if (four && regen < 6) return -ZE_CORRUPTION;
    if (hdr + csize > srcSize) return -ZE_CORRUPTION;
    const uint8_t* p = src + hdr; uint32_t left = csize;
    if (type == 3) {
        if (st.hufCount == 0) return -ZE_DICT_CORRUPTED;
    } else {
        uint32_t cnt = 0, lg = 0;
        int r = zd_read_huf_weights(L, p, left, &cnt, &lg, defer ? defer->pre : nullptr, hdr);
        if (r < 0) return -ZE_CORRUPTION;
        p += r; left -= (uint32_t)r; st.hufCount = cnt;
    }
    if (type == 3 && defer && defer->prevTable) {                  // treeless block of a dictionary frame: the ready-made table, no build
        const int lgp = (int)defer->prevLog;
        if (four) {
            if (left < 10) return -ZE_CORRUPTION;
            if (3 * ((regen + 3) / 4) > regen) return -ZE_CORRUPTION;
        }
        if (defer->shareOK) defer->shared = 1;
        else {
            const uint64_t* t8 = (const uint64_t*)defer->prevTable; uint64_t* g8 = (uint64_t*)defer->table;
            for (uint32_t i = lane; i < (1u << lgp) / 4 + ((1u << lgp) < 4 ? 1u : 0u); i += 64) g8[i] = t8[i];
        }
        defer->taken = 1; defer->log = (uint32_t)lgp; defer->four = four; defer->streamBytes = left; defer->streams = p;
        ZD_T(P, ZP_HUFTAB);
        st.litPtr = lit; st.litSize = regen;
        return (int)(hdr + csize);
    }
    int lg = zd_build_huf(L, st.hufCount);
    if (lg < 0) return -ZE_CORRUPTION;
    if (defer && (uint32_t)lg <= defer->maxLog) {
        if (four) {
            if (left < 10) return -ZE_CORRUPTION;
            if (3 * ((regen + 3) / 4) > regen) return -ZE_CORRUPTION;
        }
        const uint64_t* t8 = (const uint64_t*)L.u.huf; uint64_t* g8 = (uint64_t*)defer->table;
        for (uint32_t i = lane; i < (1u << lg) / 4 + ((1u << lg) < 4 ? 1u : 0u); i += 64) g8[i] = t8[i];
        defer->taken = 1; defer->log = (uint32_t)lg; defer->four = four; defer->streamBytes = left; defer->streams = p;
        ZD_T(P, ZP_HUFTAB);
        st.litPtr = lit; st.litSize = regen;
        return (int)(hdr + csize);
    }
    ZD_T(P, ZP_HUFTAB);
    bool ok = true;
    if (!four) {
        if (lane == 0) ok = zd_huf_stream(L.u.huf, (uint32_t)lg, p, left, lit, regen);
    } else {
        if (left < 10) return -ZE_CORRUPTION;
        uint32_t s1 = zh_ld16(p), s2 = zh_ld16(p + 2), s3 = zh_ld16(p + 4);
        if (6 + s1 + s2 + s3 > left) return -ZE_CORRUPTION;
        uint32_t s4 = left - 6 - s1 - s2 - s3;
        uint32_t seg = (regen + 3) / 4;
        if (3 * seg > regen) return -ZE_CORRUPTION;
        if (lane < 4) {
            uint32_t off = lane == 0 ? 0 : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
            uint32_t sz = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : s4;
            uint32_t n = lane < 3 ? seg : regen - 3 * seg;
            ok = zd_huf_stream(L.u.huf, (uint32_t)lg, p + 6 + off, sz, lit + lane * seg, n);
        }
    }
    if (zh_ballot(!ok)) return -ZE_CORRUPTION;
    zd_fence();
    ZD_T(P, ZP_HUFDEC);
    st.litPtr = lit; st.litSize = regen;
    return (int)(hdr + csize);
}

// ------------------------------------------------------------------------------------------ sequence tables
// one of the three symbol-compression modes (RFC 8878 3.1.1.3.2.1). All lanes call. Returns bytes used or -err.
// `pre`: K0's record of the frame or null -- a distribution it parsed is taken from there (zp_block_tables hands the record over only for the block it was made from)
ZH_DEVFN int zd_seq_table(ZdLDS& L, uint32_t mode, int kind, uint32_t* pLog, const uint8_t* p, const uint8_t* end, const ZpPre* pre = nullptr)
{
    const uint32_t lane = zh_lane();
    uint32_t* table = L.fse + (kind == ZD_KIND_LL ? ZD_FSE_LL : kind == ZD_KIND_ML ? ZD_FSE_ML : ZD_FSE_OF);
    const uint32_t maxSym = kind == ZD_KIND_LL ? ZF_MAXLL : kind == ZD_KIND_ML ? ZF_MAXML : ZF_MAXOFF;
    const uint32_t maxLog = kind == ZD_KIND_LL ? ZF_LL_LOGMAX : kind == ZD_KIND_ML ? ZF_ML_LOGMAX : ZF_OF_LOGMAX;
    if (mode == 0) {
        zh_sync();
        if (kind == ZD_KIND_LL) L.norm[lane] = lane < 36 ? zc_llDef[lane] : (int16_t)0;
        else if (kind == ZD_KIND_ML) L.norm[lane] = lane < 53 ? zc_mlDef[lane] : (int16_t)0;
        else L.norm[lane] = lane < 29 ? zc_ofDef[lane] : (int16_t)0;
        uint32_t lg = kind == ZD_KIND_OF ? 5 : 6;
        uint32_t ms = kind == ZD_KIND_LL ? 35 : kind == ZD_KIND_ML ? 52 : 28;
        if (zd_build_fse(L, table, ms, lg, kind) < 0) return -ZE_CORRUPTION;
        *pLog = lg; return 0;
    }
    if (mode == 1) {
        if (p >= end) return -ZE_SRC_SIZE_WRONG;
        uint32_t s = p[0];
        if (s > maxSym) return -ZE_CORRUPTION;
        zh_sync();
        if (lane == 0) {
            uint32_t extra = kind == ZD_KIND_LL ? L.llBits[s] : kind == ZD_KIND_ML ? L.mlBits[s] : s;
            table[0] = ZD_CELL(0, 0, extra, s);
        }
        zh_sync();
        *pLog = 0; return 1;
    }
    if (mode == 2) {
        zh_sync();
        int r; uint32_t ms, tl;
        if (pre && pre->t[kind].valid) {
            L.norm[lane] = pre->norm[kind][lane];
            r = (int)pre->t[kind].used; ms = pre->t[kind].maxSym; tl = pre->t[kind].log;
#ifdef ZHIP_EMU
            if (lane == 0) zd_stat[11]++;                                                // (test hook [11]: sequence distributions taken from K0's record)
#endif
        } else {
            if (lane == 0) {
                uint32_t ms0 = maxSym, tl0 = 0;
                int r0 = zd_read_ncount(L, p, end, &ms0, &tl0);
                L.misc[0] = (uint32_t)r0; L.misc[1] = ms0; L.misc[2] = tl0;
            }
            zh_sync();
            r = (int)L.misc[0]; ms = L.misc[1]; tl = L.misc[2];
        }
        if (r < 0 || tl > maxLog) return -ZE_CORRUPTION;
        if (zd_build_fse(L, table, ms, tl, kind) < 0) return -ZE_CORRUPTION;
        *pLog = tl; return r;
    }
    if (*pLog == 0xFF) return -ZE_CORRUPTION;    // repeat without a previous table
    return 0;
}

// byte at frame-relative position pos (negative: dictionary content)
ZH_DEV uint32_t zd_hist_byte(const uint8_t* dst, const uint8_t* dictEnd, int32_t pos)
{
    return pos >= 0 ? dst[pos] : dictEnd[pos];
}

// whole-wave match copy straight in global memory, for matches too long for the LDS assembly buffer.
// All lanes call with uniform arguments.
ZH_COLD void zd_match_wave(uint8_t* dst, const uint8_t* dictEnd, uint32_t mdst, uint32_t off, uint32_t ml)
{
    const uint32_t lane = zh_lane();
    const int32_t sbeg = (int32_t)mdst - (int32_t)off;
    if (off >= 64) {
        for (uint32_t c = 0; c < ml; c += 64) {
            uint32_t j = c + lane;
            if (j < ml) dst[mdst + j] = (uint8_t)zd_hist_byte(dst, dictEnd, sbeg + (int32_t)j);
            if (off < ml) zd_fence();       // later chunks read what this chunk wrote
        }
    } else {                                 // period `off` pattern: every source byte precedes mdst
        uint32_t idx = lane % off, adv = 64 % off;
        for (uint32_t j = lane; j < ml; j += 64) {
            dst[mdst + j] = (uint8_t)zd_hist_byte(dst, dictEnd, sbeg + (int32_t)idx);
            idx += adv; if (idx >= off) idx -= off;
        }
    }
}

// ---- up-to-32-byte moves through 4 registers: all loads are issued before any store (one memory latency per move).
// Branch-lean: full 8-byte words for the body, and for the tail either an overlapping 8-byte word ending exactly at
// `len` (len >= 8) or a 4/2/1 cascade (len < 8) -- no byte loops, since a divergent loop costs its longest lane.
// r[0..2] hold bytes [0,8) [8,16) [16,24); r[3] holds the LAST 8 bytes (overlapping) when len >= 8, else the whole run.
ZH_DEV void zd_ld32(const uint8_t* p, uint32_t len, uint64_t r[4])
{
    // every piece is a predicated load into its own register and nothing is combined before all of them are issued: written as
    // nested branches the pieces of a short run (4 + 2 + 1 bytes) were three dependent memory round trips
    const bool ge8 = len >= 8;
    const uint32_t o2 = len & 4, o1 = o2 + (len & 2);
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0; uint32_t b4 = 0, b2 = 0, b1 = 0;
    if (ge8) a0 = zh_ld64(p);
    if (len > 16) a1 = zh_ld64(p + 8);
    if (len > 24) a2 = zh_ld64(p + 16);
    if (ge8) a3 = zh_ld64(p + len - 8);
    if (!ge8 && (len & 4)) b4 = zh_ld32(p);
    if (!ge8 && (len & 2)) b2 = zh_ld16(p + o2);
    if (!ge8 && (len & 1)) b1 = p[o1];
    r[0] = a0; r[1] = a1; r[2] = a2;
    r[3] = ge8 ? a3 : ((uint64_t)b4 | ((uint64_t)b2 << (8 * o2)) | ((uint64_t)b1 << (8 * o1)));
}
ZH_DEV void zd_st32(uint8_t* q, uint32_t len, const uint64_t r[4])
{
    if (len >= 8) {
        zh_st64(q, r[0]);
        if (len > 16) zh_st64(q + 8, r[1]);
        if (len > 24) zh_st64(q + 16, r[2]);
        zh_st64(q + len - 8, r[3]);
    } else {
        uint64_t v = r[3]; uint32_t o = 0;
        if (len & 4) { zh_st32(q, (uint32_t)v); v >>= 32; o = 4; }
        if (len & 2) { zh_st16(q + o, (uint16_t)v); v >>= 16; o += 2; }
        if (len & 1) { q[o] = (uint8_t)v; }
    }
}

struct ZdPack16 { uint32_t a, b, c, d; } __attribute__((packed, aligned(1)));

// Sequences section + execution for one compressed block. All lanes call. Returns 0 or -err; *pOp advanced.
//
// Execution model: the output of one batch of <=64 sequences (<= ZD_ASM_BYTES bytes) is ASSEMBLED IN LDS and then
// flushed to HBM with wide coalesced stores. Everything a batch reads from global memory (its literals, and matches
// whose source lies before the batch) is independent of the batch itself, so those loads are issued together and
// cost one memory latency; matches that read the batch's own output are resolved in dependency rounds at LDS
// latency. Sequences too large for the buffer take the direct global path.
ZH_DEVFN int zd_sequences(ZdLDS& L, ZdState& st, const uint8_t* p, const uint8_t* end, uint8_t* dst, uint32_t cap,
                          uint32_t* pOp, uint32_t blockMax, const uint8_t* dictEnd, uint32_t dictSize, ZdProf& P)
{
    const uint32_t lane = zh_lane();
    const uint32_t blockStart = *pOp;
    uint32_t op = *pOp, lp = 0;
    if (p >= end) return -ZE_SRC_SIZE_WRONG;
    uint32_t nbSeq = *p++;
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (p + 2 > end) return -ZE_SRC_SIZE_WRONG; nbSeq = zh_ld16(p) + 0x7F00; p += 2; }
        else { if (p >= end) return -ZE_SRC_SIZE_WRONG; nbSeq = ((nbSeq - 128) << 8) + *p++; }
    }
    if (nbSeq == 0) {
        if (p != end) return -ZE_CORRUPTION;
    } else {
        if (p >= end) return -ZE_SRC_SIZE_WRONG;
        uint32_t modes = *p++;
        if (modes & 3) return -ZE_CORRUPTION;
        int r = zd_seq_table(L, modes >> 6, ZD_KIND_LL, &st.llLog, p, end); if (r < 0) return r; p += r;
        r = zd_seq_table(L, (modes >> 4) & 3, ZD_KIND_OF, &st.ofLog, p, end); if (r < 0) return r; p += r;
        r = zd_seq_table(L, (modes >> 2) & 3, ZD_KIND_ML, &st.mlLog, p, end); if (r < 0) return r; p += r;
        if (p >= end) return -ZE_CORRUPTION;
        ZD_T(P, ZP_SEQTAB);
        const uint32_t ssize = (uint32_t)(end - p);
        const uint32_t last = p[ssize - 1];
        if (last == 0) return -ZE_CORRUPTION;
        // lane-0 bit window: top V bits of w are valid; q = stream byte position of the window's lower edge
        uint64_t w; uint32_t V; int32_t q;
        {
            uint32_t pad = 8 - (uint32_t)zh_highbit32(last);
            if (ssize >= 8) { w = zh_ld64(p + ssize - 8) << pad; V = 64 - pad; q = (int32_t)ssize - 8; }
            else {
                uint64_t c = 0;
                for (uint32_t i = 0; i < ssize; i++) c |= (uint64_t)p[i] << (8 * i);
                V = 8 * ssize - pad; w = V ? c << (64 - V) : 0; q = 0;
            }
        }
        uint32_t sL = 0, sM = 0, sO = 0, done = 0;
        uint32_t rep0 = st.rep0, rep1 = st.rep1, rep2 = st.rep2;
        uint32_t carry = 0, cLL = 0, cML = 0, cOF = 0;          // lane 0: a decoded sequence that did not fit the last batch
        bool first = true;
        uint8_t* const asmb = L.u.a.asmb;
        while (done < nbSeq) {
            // ---- stage the next <=512 bytes of the backward stream into LDS (coalesced 8-byte loads)
            zh_sync();
            const int32_t stageLo = q > ZD_STAGE_BYTES ? q - ZD_STAGE_BYTES : 0;
            {
                int32_t pos = stageLo + (int32_t)lane * 8;
                if (pos < q) {
                    uint64_t v = zh_ld64(p + pos);
                    L.u.q.stage[1 + lane * 2] = (uint32_t)v;
                    L.u.q.stage[2 + lane * 2] = (uint32_t)(v >> 32);
                }
                if (zh_opaque(lane) == 0) L.u.q.stage[0] = 0;
            }
            zh_sync();
            ZD_T(P, ZP_STAGE);
            // ---- lane 0: decode up to 64 sequences / ZD_ASM_BYTES of output
            if (zh_opaque(lane) == 0) {
#define ZD_REFILL() do { uint32_t x_ = 0; if (q > 0) { uint32_t ix_ = (uint32_t)(q - stageLo); \
                        uint64_t t_ = ((uint64_t)L.u.q.stage[(ix_ >> 2) + 1] << 32) | L.u.q.stage[ix_ >> 2]; \
                        x_ = (uint32_t)(t_ >> (8 * (ix_ & 3))); } \
                        w |= (uint64_t)x_ << (32 - V); V += 32; q -= 4; } while (0)
#define ZD_TAKE(dstv, n) do { uint32_t n_ = (n); if (V < n_) ZD_REFILL(); \
                        dstv = (uint32_t)((w >> 1) >> (63 - n_)); w <<= n_; V -= n_; } while (0)
                if (first) {
                    ZD_TAKE(sL, st.llLog); ZD_TAKE(sO, st.ofLog); ZD_TAKE(sM, st.mlLog);
                }
                uint32_t cnt = 0, outAcc = 0, big = 0;
                int bad = 0;
                if (carry) {
                    L.u.q.ll[0] = cLL; L.u.q.ml[0] = cML; L.u.q.of[0] = cOF;
                    outAcc = cLL + cML; cnt = 1; carry = 0;
                    if (outAcc > ZD_ASM_BYTES) big = 1;
                }
                while (!big && cnt < 64 && done + cnt < nbSeq && (stageLo == 0 || q - stageLo >= 16)) {
                    uint32_t eL = L.fse[ZD_FSE_LL + sL], eM = L.fse[ZD_FSE_ML + sM], eO = L.fse[ZD_FSE_OF + sO];
                    uint32_t ofBits = (eO >> 14) & 31, mlBits = (eM >> 14) & 31, llBits = (eL >> 14) & 31;
                    uint32_t xo, xm, xl;
                    ZD_TAKE(xo, ofBits); ZD_TAKE(xm, mlBits); ZD_TAKE(xl, llBits);
                    uint32_t ofv = (1u << ofBits) + xo;
                    uint32_t mlv = L.mlBase[eM >> 19] + xm;
                    uint32_t llv = L.llBase[eL >> 19] + xl;
                    uint32_t offset;
                    if (ofv > 3) { offset = ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                    else {
                        uint32_t idx = ofv - 1 + (llv == 0);
                        if (idx == 0) offset = rep0;
                        else {
                            offset = idx == 3 ? rep0 - 1 : (idx == 1 ? rep1 : rep2);
                            if (offset == 0) offset = 0xFFFFFFFFu;          // rep0 - 1 == 0: no such offset -- libzstd 1.5.7 forces -1 and the execution refuses it (zstd.c:46941)
                            if (idx != 1) rep2 = rep1;
                            rep1 = rep0; rep0 = offset;
                        }
                    }
                    if (done + cnt + 1 < nbSeq) {
                        uint32_t t;
                        ZD_TAKE(t, (eL >> 10) & 15); sL = (eL & 1023) + t;
                        ZD_TAKE(t, (eM >> 10) & 15); sM = (eM & 1023) + t;
                        ZD_TAKE(t, (eO >> 10) & 15); sO = (eO & 1023) + t;
                    }
                    if ((int64_t)q * 8 + (int64_t)V < 0) { bad = 1; break; }
                    if (outAcc + llv + mlv > ZD_ASM_BYTES) {
                        if (cnt == 0) { L.u.q.ll[0] = llv; L.u.q.ml[0] = mlv; L.u.q.of[0] = offset; cnt = 1; big = 1; }
                        else { carry = 1; cLL = llv; cML = mlv; cOF = offset; }
                        break;
                    }
                    L.u.q.ll[cnt] = llv; L.u.q.ml[cnt] = mlv; L.u.q.of[cnt] = offset;
                    outAcc += llv + mlv;
                    cnt++;
                }
                // a carried sequence is not counted as done yet, so the stream-position bookkeeping stays simple
                L.misc[0] = cnt; L.misc[1] = (uint32_t)bad; L.misc[2] = (uint32_t)q; L.misc[3] = V;
                L.misc[4] = rep0; L.misc[5] = rep1; L.misc[6] = rep2; L.misc[7] = big | (carry << 1);
#undef ZD_TAKE
#undef ZD_REFILL
            }
            first = false;
            zh_sync();
            ZD_T(P, ZP_SEQDEC);
            const uint32_t cnt = L.misc[0];
            if (L.misc[1]) return -ZE_CORRUPTION;
            q = (int32_t)zh_first(L.misc[2]);
            const uint32_t big = L.misc[7] & 1, carried = L.misc[7] >> 1;
            if (cnt == 0) { if (carried) return -ZE_CORRUPTION; continue; }   // window had run dry: restage lower and retry
            const bool act = lane < cnt;
            const uint32_t myLL = act ? L.u.q.ll[lane] : 0, myML = act ? L.u.q.ml[lane] : 0, myOF = act ? L.u.q.of[lane] : 1;
            const uint32_t incL = zh_scan_add(myLL), incT = zh_scan_add(myLL + myML);
            const uint32_t totL = zh_shfl(incL, 63), totT = zh_shfl(incT, 63);
            if (lp + totL > st.litSize) return -ZE_CORRUPTION;
            if ((uint64_t)op + totT > cap) return -ZE_DST_TOO_SMALL;
            if (op + totT - blockStart > blockMax) return -ZE_CORRUPTION;
            const uint32_t litStart = lp + incL - myLL;
            const uint32_t oRel = incT - (myLL + myML);          // batch-relative output start of my sequence
            const uint32_t mRel = oRel + myLL;                   // batch-relative start of my match
            if (zh_ballot(act && (uint64_t)myOF > (uint64_t)op + mRel + dictSize)) return -ZE_CORRUPTION;
            if (big) {
                // one sequence larger than the assembly buffer: straight global copies
                const uint32_t bll = zh_shfl(myLL, 0), bml = zh_shfl(myML, 0), bof = zh_shfl(myOF, 0);
                if (st.litRLE) zd_fill_wave(dst + op, st.rleByte, bll);
                else zd_copy_wave(dst + op, st.litPtr + lp, bll);
                zd_fence();
                zd_match_wave(dst, dictEnd, op + bll, bof, bml);
                zd_fence();
                op += totT; lp += totL; done += cnt;
                continue;
            }
            // ---- phase 1: everything that comes from global memory (literals + matches older than this batch)
            const int32_t sAbs = (int32_t)(op + mRel) - (int32_t)myOF;       // frame-relative match source (negative: dictionary)
            const bool hasM = act && myML > 0;
            const bool farM = hasM && sAbs + (int32_t)myML <= (int32_t)op && (sAbs >= 0 || sAbs + (int32_t)myML <= 0);
            {
                uint64_t rl[4], rm[4];
                const bool shortL = act && myLL > 0 && myLL <= ZD_COOP_LEN;
                const bool shortFar = farM && myML <= ZD_COOP_LEN;
                if (shortL && !st.litRLE) zd_ld32(st.litPtr + litStart, myLL, rl);
                if (shortFar) zd_ld32(sAbs >= 0 ? dst + sAbs : dictEnd + sAbs, myML, rm);
                if (shortL) {
                    if (st.litRLE) { for (int k = 0; k < 4; k++) rl[k] = 0x0101010101010101ull * st.rleByte; }
                    zd_st32(asmb + oRel, myLL, rl);
                }
                if (shortFar) zd_st32(asmb + mRel, myML, rm);
            }
            for (uint64_t m = zh_ballot(act && myLL > ZD_COOP_LEN); m; m &= m - 1) {       // long literal runs: whole wave
                const uint32_t l = (uint32_t)zh_ctz64(m);
                const uint32_t d = zh_shfl(oRel, l), s = zh_shfl(litStart, l), n = zh_shfl(myLL, l);
                if (st.litRLE) { for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = (uint8_t)st.rleByte; }
                else zd_stage_wave(asmb + d, st.litPtr + s, n);
            }
            for (uint64_t m = zh_ballot(farM && myML > ZD_COOP_LEN); m; m &= m - 1) {      // long far matches: whole wave
                const uint32_t l = (uint32_t)zh_ctz64(m);
                const uint32_t d = zh_shfl(mRel, l), n = zh_shfl(myML, l);
                const int32_t s = (int32_t)zh_shfl((uint32_t)sAbs, l);
                for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = (uint8_t)zd_hist_byte(dst, dictEnd, s + (int32_t)j);
            }
            zh_sync();
            ZD_T(P, ZP_EXEC1);
            // ---- phase 2: matches that read this batch's own output, in dependency rounds at LDS speed
            bool pending = hasM && !farM;
            int32_t send = sAbs + (int32_t)myML; if (send > (int32_t)(op + mRel)) send = (int32_t)(op + mRel);
            for (;;) {
                const uint64_t pend = zh_ballot(pending);
                if (!pend) break;
                const uint32_t f = (uint32_t)zh_ctz64(pend);
                const uint32_t Frel = zh_shfl(mRel, f), fml = zh_shfl(myML, f), fof = zh_shfl(myOF, f);
                if (fml > ZD_COOP_LEN) {
                    // whole wave copies one long match; every source byte is final (older than Frel)
                    const int32_t fs = (int32_t)(op + Frel) - (int32_t)fof;
                    if (fof >= 64) {
                        for (uint32_t c = 0; c < fml; c += 64) {
                            const uint32_t j = c + lane;
                            if (j < fml) {
                                const int32_t sp = fs + (int32_t)j;
                                asmb[Frel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : (uint8_t)zd_hist_byte(dst, dictEnd, sp);
                            }
                            if (fof < fml) zh_sync();
                        }
                    } else {
                        uint32_t idx = lane % fof; const uint32_t adv = 64 % fof;
                        for (uint32_t j = lane; j < fml; j += 64) {
                            const int32_t sp = fs + (int32_t)idx;
                            asmb[Frel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : (uint8_t)zd_hist_byte(dst, dictEnd, sp);
                            idx += adv; if (idx >= fof) idx -= fof;
                        }
                    }
                    if (lane == f) pending = false;
                } else {
                    const bool ready = pending && myML <= ZD_COOP_LEN && (lane == f || send <= (int32_t)(op + Frel));
                    if (ready) {
                        if (sAbs >= (int32_t)op && myOF >= myML) {            // entirely inside the buffer, no self-overlap
                            uint64_t rr[4];
                            zd_ld32(asmb + (sAbs - (int32_t)op), myML, rr);
                            zd_st32(asmb + mRel, myML, rr);
                        } else {                                               // self-overlapping or straddling: byte serial
                            for (uint32_t j = 0; j < myML; j++) {
                                const int32_t sp = sAbs + (int32_t)j;
                                asmb[mRel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : (uint8_t)zd_hist_byte(dst, dictEnd, sp);
                            }
                        }
                        pending = false;
                    }
                }
                zh_sync();
            }
            ZD_T(P, ZP_EXEC2);
            // ---- phase 3: flush the assembled bytes to HBM, 16 bytes per lane per store
            {
                uint8_t* out = dst + op;
                for (uint32_t j = lane * 16; j < totT; j += 1024) {
                    if (j + 16 <= totT) {
                        const uint32_t* s4 = (const uint32_t*)(asmb + j);
                        ZdPack16 v; v.a = s4[0]; v.b = s4[1]; v.c = s4[2]; v.d = s4[3];
                        *(ZdPack16*)(out + j) = v;
                    } else {
                        for (uint32_t k = j; k < totT; k++) out[k] = asmb[k];
                    }
                }
            }
            op += totT; lp += totL; done += cnt;
            ZD_T(P, ZP_FLUSH);
        }
        // the bitstream must be consumed exactly
        {
            uint32_t Vu = zh_first(L.misc[3]);
            if ((int64_t)q * 8 + (int64_t)Vu != 0) return -ZE_CORRUPTION;
        }
        st.rep0 = zh_first(L.misc[4]); st.rep1 = zh_first(L.misc[5]); st.rep2 = zh_first(L.misc[6]);
        zh_sync();
    }
    // last literals
    uint32_t rest = st.litSize - lp;
    if ((uint64_t)op + rest > cap) return -ZE_DST_TOO_SMALL;
    if (op + rest - blockStart > blockMax) return -ZE_CORRUPTION;
    if (st.litRLE) zd_fill_wave(dst + op, st.rleByte, rest);
    else zd_copy_wave(dst + op, st.litPtr + lp, rest);
    zd_fence();
    ZD_T(P, ZP_RAW);
    *pOp = op + rest;
    return 0;
}

// ------------------------------------------------------------------------------------------ frame
ZH_DEVFN int zd_frame(const ZhipDecodeArgs& a, ZdLDS& L, uint32_t f, uint8_t* lit, uint64_t* produced, ZdProf& P)
{
    const uint32_t lane = zh_lane();
#if defined(ZHIP_EMU) && defined(ZD_TRACE)
    zd_cur_frame = f;
#endif
    *produced = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    if (srcSize64 > 0x7FFFFFFFull) return ZE_PARAM_UNSUPPORTED;
    const uint32_t srcSize = (uint32_t)srcSize64;
    const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
    // ---- frame header (RFC 8878 3.1.1.1)
    const uint32_t mg = a.magicless ? 0u : 4u;             // ZSTD_f_zstd1_magicless: the frame starts at its descriptor byte
    if (srcSize < mg + 1) return ZE_SRC_SIZE_WRONG;
    if (mg && (zh_ld32(src) & 0xFFFFFFF0u) == ZF_MAGIC_SKIPPABLE)        // a skippable frame: passed over, nothing produced (zstd.c:43706, :44731)
        return srcSize < 8 || (uint64_t)zh_ld32(src + 4) + 8 > srcSize ? ZE_SRC_SIZE_WRONG : ZE_OK;
    if (mg && zh_ld32(src) != ZF_MAGIC) return ZE_PREFIX_UNKNOWN;
    const uint32_t fhd = src[mg];
    const uint32_t dictCode = fhd & 3, hasChecksum = (fhd >> 2) & 1, single = (fhd >> 5) & 1, fcsCode = fhd >> 6;
    const uint32_t dictBytes = dictCode == 3 ? 4 : dictCode;
    const uint32_t fcsBytes = fcsCode == 0 ? single : (1u << fcsCode);
    const uint32_t hs = mg + 1 + (single ? 0 : 1) + dictBytes + fcsBytes;
    if (fhd & 8) return ZE_FRAMEPARAM_UNSUPPORTED;
    if (srcSize < hs) return ZE_SRC_SIZE_WRONG;
    uint32_t pos = mg + 1;
    uint64_t windowSize = 0;
    if (!single) {
        uint32_t wd = src[pos++], wl = 10 + (wd >> 3);
        if (wl > 31) return ZE_WINDOW_TOO_LARGE;
        windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (wd & 7);
    }
    uint32_t dictID = 0;
    if (dictCode == 1) dictID = src[pos]; else if (dictCode == 2) dictID = zh_ld16(src + pos); else if (dictCode == 3) dictID = zh_ld32(src + pos);
    pos += dictBytes;
    uint64_t fcs = ~0ull;
    if (fcsCode == 0) { if (single) fcs = src[pos]; }
    else if (fcsCode == 1) fcs = (uint64_t)zh_ld16(src + pos) + 256;
    else if (fcsCode == 2) fcs = zh_ld32(src + pos);
    else fcs = zh_ld64(src + pos);
    pos += fcsBytes;
    if (single) windowSize = fcs;
    if (windowSize > a.maxWindowSize) return ZE_WINDOW_TOO_LARGE;
    const uint32_t blockMax = windowSize < ZF_BLOCK_MAX ? (uint32_t)windowSize : ZF_BLOCK_MAX;
    if (dictID && dictID != a.dictID) return ZE_DICT_WRONG;
    // ---- entropy / repcode start state (fresh, or preloaded from the dictionary)
    ZdState st;
    st.rep0 = 1; st.rep1 = 4; st.rep2 = 8;
    st.hufCount = 0; st.llLog = st.ofLog = st.mlLog = 0xFF;
    st.litPtr = lit; st.litSize = 0; st.litRLE = 0; st.rleByte = 0;
    const uint8_t* dictEnd = a.dictContent ? a.dictContent + a.dictContentSize : dst;
    const uint32_t dictSize = a.dictContent ? a.dictContentSize : 0;
    if (a.dictEntropy && a.dictEntropy->hufCount) {
        const ZhipDictEntropy* de = a.dictEntropy;
        zh_sync();
        for (uint32_t i = lane; i < 256; i += 64) L.weights[i] = de->hufWeights[i];
        st.hufCount = de->hufCount;
        L.norm[lane] = lane < 36 ? de->llNorm[lane] : (int16_t)0;
        if (zd_build_fse(L, L.fse + ZD_FSE_LL, de->llMax, de->llLog, ZD_KIND_LL) < 0) return ZE_DICT_CORRUPTED;
        zh_sync();
        L.norm[lane] = lane < 32 ? de->ofNorm[lane] : (int16_t)0;
        if (zd_build_fse(L, L.fse + ZD_FSE_OF, de->ofMax, de->ofLog, ZD_KIND_OF) < 0) return ZE_DICT_CORRUPTED;
        zh_sync();
        L.norm[lane] = lane < 53 ? de->mlNorm[lane] : (int16_t)0;
        if (zd_build_fse(L, L.fse + ZD_FSE_ML, de->mlMax, de->mlLog, ZD_KIND_ML) < 0) return ZE_DICT_CORRUPTED;
        st.llLog = de->llLog; st.ofLog = de->ofLog; st.mlLog = de->mlLog;
        st.rep0 = de->rep[0]; st.rep1 = de->rep[1]; st.rep2 = de->rep[2];
    }
    // ---- blocks
    // The reference hands the frame to ZSTD_decompressStream with an output of the declared size (c-ext/decompressor.c:1150). With the content
    // size in the header and room for it libzstd decodes in one pass (zstd.c:44174): raw and RLE blocks of any size pass, a compressed block
    // above the frame's block maximum is srcSize_wrong (zstd.c:47714). Otherwise it streams and any block above the maximum is corruption.
    const bool onePass = fcs != ~0ull && cap64 >= fcs;
    uint32_t op = 0;
    for (;;) {
        if (pos + 3 > srcSize) return ZE_SRC_SIZE_WRONG;
        const uint32_t bh = zh_ld24(src + pos); pos += 3;
        const uint32_t lastBlock = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
        if (type == 3) return ZE_CORRUPTION;
        if (type == 0) {
            if (pos + bs > srcSize) return ZE_SRC_SIZE_WRONG;
            if (bs > blockMax && !onePass) return ZE_CORRUPTION;
            if ((uint64_t)op + bs > cap) return ZE_DST_TOO_SMALL;
            zd_copy_wave(dst + op, src + pos, bs);
            zd_fence();
            ZD_T(P, ZP_RAW);
            op += bs; pos += bs;
        } else if (type == 1) {
            if (pos + 1 > srcSize) return ZE_SRC_SIZE_WRONG;
            if (bs > blockMax && !onePass) return ZE_CORRUPTION;
            if ((uint64_t)op + bs > cap) return ZE_DST_TOO_SMALL;
            zd_fill_wave(dst + op, src[pos], bs);
            zd_fence();
            op += bs; pos += 1;
        } else {
            if (pos + bs > srcSize) return ZE_SRC_SIZE_WRONG;
            if (bs > blockMax) return onePass ? ZE_SRC_SIZE_WRONG : ZE_CORRUPTION;
            if (bs < 2) return ZE_CORRUPTION;
            ZD_T(P, ZP_HEADER);
            int r = zd_literals(L, st, src + pos, bs, lit, blockMax, P);
            if (r < 0) return -r;
            int e = zd_sequences(L, st, src + pos + r, src + pos + bs, dst, cap, &op, blockMax, dictEnd, dictSize, P);
            if (e < 0) return -e;
            pos += bs;
        }
        if (lastBlock) break;
    }
    if (fcs != ~0ull && fcs != op) return ZE_CORRUPTION;
    if (hasChecksum) {
        // content checksum (zstd.c:44270-44277): low 32 bits of XXH64 over everything this frame produced
        if (pos + 4 > srcSize) return ZE_CHECKSUM_WRONG;
        zd_fence();
        zh_sync();
        if (zh_opaque(lane) == 0) L.misc[0] = (uint32_t)ze_xxh64(dst, op);
        zh_sync();
        const uint32_t digest = zh_first(L.misc[0]);
        zh_sync();
        if (digest != zh_ld32(src + pos)) return ZE_CHECKSUM_WRONG;
    }
    *produced = op;
    return ZE_OK;
}

ZH_DEVFN void zd_kernel_body(const ZhipDecodeArgs& a, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    if (lane < 36) { L.llBase[lane] = zc_llBase[lane]; L.llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { L.mlBase[lane] = zc_mlBase[lane]; L.mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    uint8_t* lit = a.scratch + (size_t)zh_block() * ZHIP_LIT_STRIDE;
    for (;;) {
        // NOTE (compiler hazard, found on hardware): never write `x = c; if (lane == 0) x = ...; x = readfirstlane(x)`.
        // LLVM threads the constant arm of that phi through the convergent readfirstlane and the non-zero lanes end up
        // in their own loop. Lane-0 results are always published through LDS + barrier instead.
        // and never put a lane-invariant branch (`if (lane == 0)`) first in a loop body: LLVM splits the back edge per
        // lane class, StructurizeCFG nests the two loops, and lanes != 0 spin forever waiting for lane 0 (SIMT deadlock).
        // The work-stealing fetch is therefore branch-free: every lane issues the add, only lane 0 adds 1.
        const uint32_t got = zh_atomic_add(a.counter, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        uint32_t f = zh_first(L.misc[7]);
        zh_sync();
        const uint32_t limit = a.listCount ? *a.listCount : a.n;
        if (f >= limit) break;
        if (a.frameList) f = a.frameList[f];
        uint64_t produced = 0;
        ZdProf P; P.on = a.prof != nullptr;
        if (P.on) { for (int i = 0; i < ZP_N; i++) P.acc[i] = 0; P.t0 = zd_clock(); }
        int err = zd_frame(a, L, f, lit, &produced, P);
        if (P.on) {
            ZD_T(P, ZP_HEADER);
            if (zh_opaque(lane) == 0) for (int i = 0; i < ZP_N; i++) zh_atomic_add64(a.prof + i, (unsigned long long)P.acc[i]);
        }
        zh_sync();
        if (zh_opaque(lane) == 0) { a.status[f] = err; a.outSizes[f] = err ? 0 : produced; }
    }
}

// Parses the entropy section of a zstd-format dictionary (magic, dictID, Huffman table, OF/ML/LL distributions,
// three repcodes -- what ZSTD_loadDEntropy, zstd.c:44673, consumes) with the same device routines the decoder uses.
// One wave. Raw-content dictionaries (no magic) yield hufCount == 0, contentOffset == 0.
ZH_DEVFN void zd_dict_body(const uint8_t* dict, uint32_t dictSize, ZhipDictEntropy* de, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    if (lane < 36) { L.llBase[lane] = zc_llBase[lane]; L.llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { L.mlBase[lane] = zc_mlBase[lane]; L.mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    int status = 0;
    uint32_t hufCount = 0, contentOffset = 0, dictID = 0;
    if (dictSize >= 8 && zh_ld32(dict) == ZF_DICT_MAGIC) {
        dictID = zh_ld32(dict + 4);
        const uint8_t* p = dict + 8; const uint8_t* end = dict + dictSize;
        uint32_t cnt = 0, lg = 0;
        int r = zd_read_huf_weights(L, p, (uint32_t)(end - p), &cnt, &lg);
        if (r < 0) status = ZE_DICT_CORRUPTED;
        else {
            p += r; hufCount = cnt;
            for (uint32_t i = lane; i < 256; i += 64) de->hufWeights[i] = i < cnt ? L.weights[i] : (uint8_t)0;
            for (int t = 0; t < 3 && !status; t++) {             // order in the dictionary: OF, ML, LL
                uint32_t lim = t == 0 ? ZF_MAXOFF : t == 1 ? ZF_MAXML : ZF_MAXLL;
                uint32_t maxLog = t == 0 ? ZF_OF_LOGMAX : ZF_ML_LOGMAX;
                zh_sync();
                if (zh_opaque(lane) == 0) {
                    uint32_t ms = lim, tl = 0;
                    int rr = zd_read_ncount(L, p, end, &ms, &tl);
                    L.misc[0] = (uint32_t)rr; L.misc[1] = ms; L.misc[2] = tl;
                }
                zh_sync();
                int rr = (int)L.misc[0]; uint32_t ms = L.misc[1], tl = L.misc[2];
                if (rr < 0 || tl > maxLog) { status = ZE_DICT_CORRUPTED; break; }
                int16_t v = L.norm[lane];
                if (t == 0) { if (lane < 32) de->ofNorm[lane] = v; de->ofMax = ms; de->ofLog = tl; }
                else if (t == 1) { if (lane < 53) de->mlNorm[lane] = v; de->mlMax = ms; de->mlLog = tl; }
                else { if (lane < 36) de->llNorm[lane] = v; de->llMax = ms; de->llLog = tl; }
                p += rr;
            }
            if (!status) {
                if (p + 12 > end) status = ZE_DICT_CORRUPTED;
                else {
                    uint32_t content = (uint32_t)(end - (p + 12));
                    for (int i = 0; i < 3; i++) {
                        uint32_t rep = zh_ld32(p + 4 * i);
                        if (rep == 0 || rep > content) status = ZE_DICT_CORRUPTED;
                        if (lane == 0) de->rep[i] = rep;
                    }
                    contentOffset = (uint32_t)(p + 12 - dict);
                }
            }
        }
    }
    if (lane == 0) { de->hufCount = status ? 0 : hufCount; de->contentOffset = contentOffset; de->dictID = dictID; de->status = status; }
}
