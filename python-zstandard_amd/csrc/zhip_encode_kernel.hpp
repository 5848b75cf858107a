// zhip_encode_kernel.hpp -- batch zstd frame encoder, one wavefront per frame, frames bit-identical to libzstd 1.5.7
// at the same level (double-fast strategy: level 3 for every size class, and the other dfast rows of the level table).
//
// Replaces the arithmetic under compress_worker() (c-ext/compressor.c:856-1076): ZSTD_compressStream2(e_end) ->
// ZSTD_compress_frameChunk -> ZSTD_compressBlock_internal (zstd.c:29401, :27545, :27337) for single-block frames
// (<= 128 KiB, every batch configuration of BASELINE.json). SURVEY.md 8(a) rows C1-C13 / Appendix A list the
// behaviours that decide output bytes; each routine below names the reference routine it restates.
//
// Round-1 shape (correctness first, see DESIGN.md for the plan): the double-fast search is a serial dependency chain
// (every step reads two hash tables the previous step wrote), so lane 0 drives it with the tables in a per-wave HBM
// workspace; the wave zeroes tables, builds histograms and symbol codes in parallel; entropy-table construction is
// small and serial; literals are Huffman-coded by four lanes (one per stream); the tANS sequence stream is emitted by
// lane 0. Lane-0 results are published through LDS + barrier (see the compiler-hazard note in zhip_decode_kernel.hpp).
#pragma once
#include "zhip_device.hpp"
#include "zhip_format.hpp"
#include "zhip_xxh64.hpp"

ZH_CONST uint32_t ze_llBase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,
                                   1024,2048,4096,8192,16384,32768,65536};
ZH_CONST uint8_t ze_llBits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
ZH_CONST uint32_t ze_mlBase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,
                                   33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
ZH_CONST uint8_t ze_mlBits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                  1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
ZH_CONST int16_t ze_llDef[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
ZH_CONST int16_t ze_mlDef[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                 1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
ZH_CONST int16_t ze_ofDef[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
struct ZeNode { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; };

struct ZeLDS {
    uint32_t hist[256];
    ZeNode node[2 * 256 + 2];   // Huffman tree nodes; once the code lengths are final the area is scratch (ze_scratch, ze_cell_sym, ze_weights)
    uint8_t hufBits[256];
    uint16_t hufCode[256];
    ZeCTab tab[3];              // LL, OF, ML
    uint32_t cnt[3][64];        // code histograms
    uint32_t misc[16];
    uint32_t stack[64];
};
// state that survives a block of a multi-block frame (advanced only when the block is emitted compressed, zstd.c:27391); only the
// generic one-wave-per-frame kernel has it
struct ZeLDSMulti {
    uint32_t mrep[2];
    uint32_t prevRepeat, prevMaxSym;    // previous block's Huffman table: 0 none, 1 usable after validation, 2 valid (a dictionary's, still unreplaced)
    uint32_t fseRep[3];                 // LL / OF / ML: 2 while the dictionary's table may still be repeated (FSE_repeat_valid), else 0 / 1
    uint8_t prevBits[256];
    uint16_t prevCode[256];
};
struct ZePrevHuf { const uint8_t* bits; const uint16_t* code; uint32_t maxSym, repeat; };   // a candidate table for the literals (dictionary or previous block)

struct ZePar { int wlog, clog, hlog, mml, strat, tlen; };
// optional per-phase cycle totals of the entropy kernel (ZHIP_PROF tuning aid; lives in registers, null when off)
#ifdef ZE_PROF_STREAM       // diagnostic build: the sequence stream's rounds split into their four parts (eight bytes of accumulator each: not in the product's registers)
enum { ZEP_GATHER = 0, ZEP_LITSTAT, ZEP_HUFBUILD, ZEP_HUFENC, ZEP_SEQSTAT, ZEP_SEQTAB, ZEP_SEQENC, ZEP_REST, ZEP_SQ_PRE, ZEP_SQ_CHAIN, ZEP_SQ_PACK, ZEP_SQ_FLUSH, ZEP_N };
#else
enum { ZEP_GATHER = 0, ZEP_LITSTAT, ZEP_HUFBUILD, ZEP_HUFENC, ZEP_SEQSTAT, ZEP_SEQTAB, ZEP_SEQENC, ZEP_REST, ZEP_N };
#endif
struct ZeProf { uint64_t t0; uint64_t acc[ZEP_N]; };
#define ZE_T(P, i) do { if (P) { const uint64_t t1_ = zd_clock(); (P)->acc[i] += t1_ - (P)->t0; (P)->t0 = t1_; } } while (0)
// scratch inside the tree-node area, valid while no tree is being built and disjoint from what ze_scratch users touch at the same time:
// the FSE table builder's symbol spread (512 B at +3072) and the Huffman weights (256 B at +3584)
ZH_DEV uint8_t* ze_cell_sym(ZeLDS& L) { return (uint8_t*)L.node + 3072; }
ZH_DEV uint8_t* ze_weights(ZeLDS& L) { return (uint8_t*)L.node + 3584; }
// small per-routine arrays of the serial table builders live here too: a private array indexed at run time would sit in scratch
// (global) memory, a ~500-cycle round trip per access inside loops that only lane 0 runs
ZH_DEV uint16_t* ze_fill_area(ZeLDS& L) { return (uint16_t*)((uint8_t*)L.node + 3840); }     // 64 x u16
ZH_DEV int16_t* ze_norm_area(ZeLDS& L) { return (int16_t*)((uint8_t*)L.node + 3968); }       // 64 x i16
static_assert(sizeof(ZeNode) * (2 * 256 + 2) >= 3968 + 128, "node area too small for its scratch uses");

// ------------------------------------------------------------------------------------------ LSB-first bit writer (one lane)
struct ZeBits { uint8_t* p; uint32_t cap; uint64_t acc; uint32_t n; uint32_t pos; };
ZH_DEV void ze_bw_init(ZeBits& b, uint8_t* p, uint32_t cap) { b.p = p; b.cap = cap; b.acc = 0; b.n = 0; b.pos = 0; }
ZH_DEV void ze_bw_add(ZeBits& b, uint32_t v, uint32_t nb)
{
    if (nb == 0) return;
    b.acc |= ((uint64_t)v & ((1ull << nb) - 1)) << b.n;
    b.n += nb;
    while (b.n >= 8) {
        if (b.pos < b.cap) b.p[b.pos] = (uint8_t)b.acc;
        b.pos++; b.acc >>= 8; b.n -= 8;
    }
}
ZH_DEV uint32_t ze_bw_close(ZeBits& b)      // end mark + padding; 0 on overflow
{
    ze_bw_add(b, 1, 1);
    if (b.n) { if (b.pos < b.cap) b.p[b.pos] = (uint8_t)b.acc; b.pos++; }
    return b.pos > b.cap ? 0 : b.pos;
}

// ------------------------------------------------------------------------------------------ FSE, compression side
// FSE_optimalTableLog_internal, zstd.c:16294
ZH_DEV uint32_t ze_fse_optimal_log(uint32_t maxLog, uint32_t n, uint32_t maxSym, uint32_t minus)
{
    uint32_t maxBitsSrc = (uint32_t)zh_highbit32(n - 1) - minus;
    uint32_t minBitsSrc = (uint32_t)zh_highbit32(n) + 1, minBitsSym = (uint32_t)zh_highbit32(maxSym) + 2;
    uint32_t minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    uint32_t lg = maxLog;
    if (maxBitsSrc < lg) lg = maxBitsSrc;
    if (minBits > lg) lg = minBits;
    if (lg < 5) lg = 5;
    if (lg > 12) lg = 12;
    return lg;
}

ZH_DEV void ze_fse_build_rle(ZeCTab& t, uint32_t sym)      // FSE_buildCTable_rle, zstd.c:16465
{
    t.log = 0; t.maxSym = sym;
    for (uint32_t s = 0; s <= sym; s++) { t.norm[s] = 0; t.cellOf[s] = 0; }
    t.norm[sym] = 1; t.cellOf[sym + 1] = 1; t.next[0] = 1;
}
// FSE_initCState2 (zstd.c:2774): the virtual start state that lands on one of the symbol's first cells
ZH_DEV uint32_t ze_fse_first_state(const ZeCTab& t, uint32_t s)
{
    if (t.log == 0) return 1;
    const uint32_t c = t.norm[s] == -1 ? 1u : (uint32_t)t.norm[s];
    const uint32_t nb = c > 1 ? (uint32_t)t.log - (uint32_t)zh_highbit32(c - 1) : (uint32_t)t.log;
    const uint32_t delta = (nb << 16) - (c << nb);
    const uint32_t nbOut = (delta + (1u << 15)) >> 16;
    const uint32_t value = (nbOut << 16) - delta;
    return t.next[t.cellOf[s] + (value >> nbOut) - c];
}
// FSE_encodeSymbol (zstd.c:2785)
ZH_DEV uint32_t ze_fse_encode(const ZeCTab& t, ZeBits& b, uint32_t v, uint32_t s)
{
    if (t.log == 0) return v;
    const uint32_t c = t.norm[s] == -1 ? 1u : (uint32_t)t.norm[s];
    uint32_t nb;
    if (c == 1) nb = (uint32_t)t.log;
    else { const uint32_t maxBits = (uint32_t)t.log - (uint32_t)zh_highbit32(c - 1); nb = v >= (c << maxBits) ? maxBits : maxBits - 1; }
    ze_bw_add(b, v, nb);
    return t.next[t.cellOf[s] + (v >> nb) - c];
}

// ------------------------------------------------------------------------------------------ FSE tables by the whole wave
// The table of an alphabet of <= 64 symbols is built with LANE s OWNING SYMBOL s: a symbol's share of the table is its own arithmetic,
// what the symbols owe each other (cells left over, who absorbs the rounding error, where a description's field starts) is a wave sum,
// a wave maximum or a prefix sum. All lanes call; arrays (count, norm, tables, scratch) are in LDS.
ZH_DEV uint32_t ze_wave_sum(uint32_t v) { return zh_shfl(zh_scan_add(v), 63); }
ZH_DEV uint64_t ze_scan_add64(uint64_t v)                           // inclusive, only on the rare normalisation fallback
{
    const uint32_t lane = zh_lane();
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t lo = zh_shfl_up((uint32_t)v, d), hi = zh_shfl_up((uint32_t)(v >> 32), d);
        if (lane >= d) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
ZH_CONST uint32_t ze_rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};     // FSE_normalizeCount's rounding thresholds (zstd.c:16405)

// The fallback distribution (FSE_normalizeM2, zstd.c:16316), entered when plain rounding would take more than half of the largest
// symbol's cells away: rare symbols get their minimum first, the rest share what is left in proportion, by running sums of
// count x rStep (each symbol's cell count is a difference of two consecutive sums -- a 64-bit prefix sum here).
ZH_COLD int ze_fse_normalize_m2_wave(int16_t* norm, uint32_t lg, uint32_t c, uint32_t total, uint32_t maxSym, int32_t lowProb)
{
    const uint32_t lane = zh_lane();
    const bool in = lane <= maxSym;
    const uint64_t lt = zh_lt_mask();
    const int32_t UNSET = -2;
    const uint32_t size = 1u << lg;
    uint32_t lowOne = (uint32_t)(((uint64_t)total * 3) >> (lg + 1));
    int32_t v = 0;
    bool given = false;
    if (in && c) {
        if (c <= (total >> lg)) { v = lowProb; given = true; }
        else if (c <= lowOne) { v = 1; given = true; }
        else v = UNSET;
    }
    uint32_t distributed = (uint32_t)zh_popc64(zh_ballot(given));
    total -= ze_wave_sum(given ? c : 0u);
    uint32_t toDistribute = size - distributed;
    if (toDistribute) {
        if (total / toDistribute > lowOne) {                       // plenty left: one more round of "small enough for a single cell"
            lowOne = (uint32_t)(((uint64_t)total * 3) / (toDistribute * 2));
            const bool more = v == UNSET && c <= lowOne;
            if (more) v = 1;
            distributed += (uint32_t)zh_popc64(zh_ballot(more));
            total -= ze_wave_sum(more ? c : 0u);
            toDistribute = size - distributed;
        }
        if (distributed == maxSym + 1) {                           // every symbol is small: the first most frequent one takes the rest
            const uint32_t best = 255u - (zh_wave_max(in ? (c << 8) | (255u - lane) : 0u) & 255u);
            if (lane == best) v += (int32_t)toDistribute;
        } else if (total == 0) {                                   // only single-cell symbols left: one more cell each, in turn, from symbol 0
            const uint64_t has = zh_ballot(v > 0);
            const uint32_t P = (uint32_t)zh_popc64(has), rank = (uint32_t)zh_popc64(has & lt);
            if (v > 0) v += (int32_t)(toDistribute / P + (rank < toDistribute % P));
        } else {
            const uint32_t vStepLog = 62 - lg;
            const uint64_t mid = (1ull << (vStepLog - 1)) - 1;
            const uint64_t rStep = (((1ull << vStepLog) * toDistribute) + mid) / total;
            const uint64_t mine = v == UNSET ? (uint64_t)c * rStep : 0ull;
            const uint64_t end = mid + ze_scan_add64(mine);
            const uint32_t weight = (uint32_t)(end >> vStepLog) - (uint32_t)((end - mine) >> vStepLog);
            if (zh_ballot(v == UNSET && weight < 1)) return -1;
            if (v == UNSET) v = (int32_t)weight;
        }
    }
    if (in) norm[lane] = (int16_t)v;
    return 0;
}
// FSE_normalizeCount (zstd.c:16402): count[] -> norm[] summing to 1 << lg. 0, or -1 when no distribution exists.
ZH_COLD int ze_fse_normalize_wave(int16_t* norm, uint32_t lg, const uint32_t* count, uint32_t total, uint32_t maxSym, bool useLowProb)
{
    const uint32_t lane = zh_lane();
    const bool in = lane <= maxSym;
    const uint32_t c = in ? count[lane] : 0u;
    const int32_t lowProb = useLowProb ? -1 : 1;
    if (zh_ballot(in && c == total)) return 0;                     // a single symbol: the callers code that as RLE before getting here
    const uint32_t scale = 62 - lg;
    const uint64_t step = (1ull << 62) / total, vStep = 1ull << (scale - 20);
    const bool low = c != 0 && c <= (total >> lg);
    int32_t p = 0;                                                   // my share, rounded down -- or up, past a threshold that depends on the share
    if (c && !low) {
        const uint64_t cs = (uint64_t)c * step;
        p = (int32_t)(cs >> scale);
        if (p < 8) p += (cs - ((uint64_t)p << scale)) > vStep * ze_rtb[p];
    }
    const int32_t still = (int32_t)(1u << lg) - (int32_t)ze_wave_sum(low ? 1u : (uint32_t)p);       // cells the rounding left over (or overdrew)
    const uint32_t key = zh_wave_max(p > 0 ? ((uint32_t)p << 8) | (255u - lane) : 0u);            // the first symbol with the largest share absorbs it
    const uint32_t largest = key ? 255u - (key & 255u) : 0u;
    const int32_t mine = !c ? 0 : low ? lowProb : p;
    const int32_t atLargest = (int32_t)zh_shfl((uint32_t)mine, largest);
    if (-still >= (atLargest >> 1)) return ze_fse_normalize_m2_wave(norm, lg, c, total, maxSym, lowProb);
    if (in) norm[lane] = (int16_t)(lane == largest ? mine + still : mine);
    return 0;
}

// The table description (FSE_writeNCount_generic, zstd.c:16170). A symbol's field is its count + 1 in as many bits as the cells still
// unassigned before it allow -- a prefix sum of the cell counts gives every lane that number directly; the zeros that follow a zero
// are counted by a run code in front of the next symbol that has cells. Every lane shifts its bits to its prefix-sum offset in a zeroed
// LDS strip; the strip is then copied out. asm32: 32 dwords of scratch. Returns bytes written.
ZH_COLD uint32_t ze_fse_write_ncount_wave(uint8_t* out, const int16_t* norm, uint32_t maxSym, uint32_t lg, uint32_t* asm32)
{
    const uint32_t lane = zh_lane();
    const bool in = lane <= maxSym;
    const int32_t n = in ? (int32_t)norm[lane] : 0;
    const uint32_t cells = n < 0 ? 1u : (uint32_t)n;
    const uint32_t remaining = (1u << lg) + 1 - (zh_scan_add(cells) - cells);      // cells (+ 1) not yet given out when my field is written
    const uint64_t zeros = zh_ballot(in && n == 0);
    const bool afterZero = lane > 0 && ((zeros >> (lane - 1)) & 1);
    // written: every symbol while cells remain, except the zeros behind a zero
    const bool written = in && remaining > 1 && !(n == 0 && afterZero);
    uint64_t bits = 0; uint32_t len = 0;
    if (lane < 32) asm32[lane] = 0;
    if (written) {
        if (afterZero) {                                                           // run code: the zeros between the run's first one and me
            const uint64_t below = ~zeros & zh_lt_mask();                         // symbols with cells below me
            const uint32_t first = below ? 64u - (uint32_t)zh_clz64(below) : 0u;  // the run's first zero (written like any symbol)
            const uint32_t z = lane - 1 - first;
            const uint32_t k24 = z / 24, k3 = (z % 24) / 3;
            bits = (k24 ? (1ull << (16 * k24)) - 1 : 0ull) | (((1ull << (2 * k3)) - 1) << (16 * k24)) | ((uint64_t)(z % 3) << (16 * k24 + 2 * k3));
            len = 16 * k24 + 2 * k3 + 2;
        }
        const uint32_t hb = (uint32_t)zh_highbit32(remaining), threshold = 1u << hb;
        const uint32_t max = 2 * threshold - 1 - remaining;
        uint32_t v = (uint32_t)(n + 1);
        if (v >= threshold) v += max;
        const uint32_t nb = hb + 1 - (v < max);
        bits |= (uint64_t)(v & ((1u << nb) - 1)) << len; len += nb;
        if (lane == 0) { bits = (bits << 4) | (lg - 5); len += 4; }
    }
    zh_sync();
    const uint32_t end = zh_scan_add(len), off = end - len;
    if (len) {
        const uint32_t w = off >> 5, sh = off & 31;                              // up to 61 bits shifted by up to 31: three dwords
        const uint64_t lo = bits << sh;
        const uint32_t top = sh ? (uint32_t)(bits >> (64 - sh)) : 0u;
        zh_lds_atomic_or(&asm32[w], (uint32_t)lo);
        if (lo >> 32) zh_lds_atomic_or(&asm32[w + 1], (uint32_t)(lo >> 32));
        if (top) zh_lds_atomic_or(&asm32[w + 2], top);
    }
    zh_sync();
    const uint32_t bytes = (zh_shfl(end, 63) + 7) >> 3;
    for (uint32_t i = lane; i < bytes; i += 64) out[i] = ((const uint8_t*)asm32)[i];
    return bytes;
}

// The encoding table (FSE_buildCTable_wksp, zstd.c:16005): the same symbol spread as the decoding side (the k-th cell handed out sits at
// (k x step) mod size, low-probability symbols take the top cells), then every symbol's cells listed in table order. One lane per symbol
// for the bookkeeping, one lane per cell for spreading and listing. cellSym: 512 bytes, ends / run: 64 x u16 each, all LDS scratch.
ZH_COLD void ze_fse_build_ctab_wave(ZeCTab& t, const int16_t* norm, uint32_t maxSym, uint32_t lg, uint8_t* cellSym, uint16_t* ends, uint16_t* run)
{
    const uint32_t lane = zh_lane();
    const uint32_t S = 1u << lg, mask = S - 1, step = (S >> 1) + (S >> 3) + 3;
    const uint64_t lt = zh_lt_mask();
    const bool in = lane <= maxSym;
    const int32_t n = in ? (int32_t)norm[lane] : 0;
    const bool low = n == -1;
    const uint64_t lowMask = zh_ballot(low);
    const uint32_t high = S - 1 - (uint32_t)zh_popc64(lowMask);
    const uint32_t cnt = n > 0 ? (uint32_t)n : 0u, cells = low ? 1u : cnt;
    const uint32_t inclCells = zh_scan_add(cells), incl = zh_scan_add(cnt);
    if (lane == 0) { t.log = (int32_t)lg; t.maxSym = maxSym; }
    if (in) { t.norm[lane] = (int16_t)n; t.cellOf[lane] = (uint16_t)(inclCells - cells); }
    if (lane == maxSym) t.cellOf[lane + 1] = (uint16_t)inclCells;
    if (low) cellSym[S - 1 - (uint32_t)zh_popc64(lowMask & lt)] = (uint8_t)lane;
    ends[lane] = (uint16_t)incl; run[lane] = 0;
    ze_fence();
    zh_sync();
    uint32_t jbase = 0;
    for (uint32_t c = 0; c < S; c += 64) {                                         // spread: a visit above `high` hands out nothing
        const uint32_t k = c + lane, p = (k * step) & mask;
        const bool valid = k < S && p <= high;
        const uint64_t m = zh_ballot(valid);
        if (valid) {
            const uint32_t j = jbase + (uint32_t)zh_popc64(m & lt);
            uint32_t pos = 0;                                                       // the symbol whose cells include the j-th one handed out
            for (uint32_t sb = 32; sb; sb >>= 1) if (ends[pos + sb - 1] <= j) pos += sb;
            cellSym[p] = (uint8_t)pos;
        }
        jbase += (uint32_t)zh_popc64(m);
    }
    ze_fence();
    zh_sync();
    for (uint32_t c = 0; c < S; c += 64) {                                         // list: cell u is the (cells of its symbol below u)-th of that symbol
        const uint32_t u = c + lane;
        const bool act = u < S;
        const uint32_t sy = act ? cellSym[u] : 0u;
        uint64_t same = zh_ballot(act);
        for (int b = 0; b < 6; b++) { const uint64_t bm = zh_ballot(((sy >> b) & 1) != 0); same &= ((sy >> b) & 1) ? bm : ~bm; }
        uint32_t r = 0;
        if (act) r = (uint32_t)run[sy] + (uint32_t)zh_popc64(same & lt);
        zh_sync();
        if (act) {
            if ((same >> lane) >> 1 == 0) run[sy] = (uint16_t)(r + 1);           // the highest lane of its group keeps the count
            t.next[t.cellOf[sy] + r] = (uint16_t)(S + u);
        }
        ze_fence();
        zh_sync();
    }
}

// ------------------------------------------------------------------------------------------ Huffman, compression side (lane 0)
ZH_DEV uint32_t ze_huf_bucket(uint32_t c) { return c < 166 ? c : (uint32_t)zh_highbit32(c) + 158; }

ZH_DEVFN void ze_huf_insertion(ZeNode* a, int n)        // HUF_insertionSort, zstd.c:17312
{
    for (int i = 1; i < n; i++) {
        ZeNode key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
ZH_DEVFN int ze_huf_partition(ZeNode* a, int low, int high)   // HUF_quickSortPartition, zstd.c:17328
{
    const uint32_t pivot = a[high].count; int i = low - 1;
    for (int j = low; j < high; j++) if (a[j].count > pivot) { i++; ZeNode t = a[i]; a[i] = a[j]; a[j] = t; }
    ZeNode t = a[i + 1]; a[i + 1] = a[high]; a[high] = t;
    return i + 1;
}
// HUF_simpleQuickSort (zstd.c:17345) without recursion: frames are (low, high, isCall). A "call" on fewer than 8
// elements is an insertion sort; a continued loop always partitions, exactly like the reference.
ZH_DEVFN void ze_huf_quicksort(ZeNode* a, int low0, int high0, uint32_t* stack)
{
    int sp = 0;
    stack[sp++] = (uint32_t)low0 | ((uint32_t)high0 << 10) | (1u << 20);
    while (sp > 0) {
        const uint32_t f = stack[--sp];
        int low = (int)(f & 1023), high = (int)((f >> 10) & 1023); const bool call = (f >> 20) & 1;
        if (high >= 1000) high -= 1024;          // -1 encoded
        if (call && high - low < 8) { ze_huf_insertion(a + low, high - low + 1); continue; }
        if (!(low < high)) continue;
        const int idx = ze_huf_partition(a, low, high);
        if (idx - low < high - idx) {
            stack[sp++] = (uint32_t)(idx + 1) | ((uint32_t)(high & 1023) << 10);                 // continue loop on the right
            stack[sp++] = (uint32_t)low | ((uint32_t)((idx - 1) & 1023) << 10) | (1u << 20);     // call on the left
        } else {
            stack[sp++] = (uint32_t)low | ((uint32_t)((idx - 1) & 1023) << 10);
            stack[sp++] = (uint32_t)(idx + 1) | ((uint32_t)(high & 1023) << 10) | (1u << 20);
        }
    }
}

// lanes of the wave that hold the same `bits`-bit key as the caller (among lanes with act set)
ZH_DEV uint64_t ze_same_key(uint32_t key, int bits, bool act)
{
    uint64_t same = zh_ballot(act);
    for (int b = 0; b < bits; b++) { const uint64_t bm = zh_ballot(((key >> b) & 1) != 0); same &= ((key >> b) & 1) ? bm : ~bm; }
    return same;
}

// Huffman code lengths and codes for the literals (HUF_buildCTable_wksp, zstd.c:17513). counts in L.hist -> L.hufBits / L.hufCode.
// All lanes call; returns the maximum code length. What is a per-symbol or per-node affair runs on all lanes -- clearing, the bucket
// histogram and its suffix sums, placing the symbols in sorted order (rank inside a bucket = symbols of the same bucket below, by
// match-any ballots), depths (every leaf walks up to the root), the per-length code numbering. What is one chain stays on lane 0:
// the quicksort of the wide buckets (its order among equal counts IS the format's tie-break, so it is the reference's algorithm,
// zstd.c:17312-17375), the two-queue merge of the tree (:17438) and the height limiter (:17133), which rarely runs.
ZH_COLD uint32_t ze_huf_build(ZeLDS& L, uint32_t maxSym, uint32_t maxBits)
{
    const uint32_t lane = zh_lane();
    const uint64_t lt = zh_lt_mask();
    ZeNode* const tbl = L.node;
    ZeNode* const node = tbl + 1;
    const uint32_t* count = L.hist;
    uint32_t* base = &L.cnt[0][0];          // 192 entries (3 x 64): symbols in this bucket or a higher one
    uint16_t* cur = L.tab[2].next;          // scratch (free until the sequence tables are built): the next free slot of every bucket
    zh_sync();
    for (uint32_t i = lane; i < 2 * 256 + 2; i += 64) { tbl[i].count = 0; tbl[i].parent = 0; tbl[i].byte = 0; tbl[i].nbBits = 0; }
    for (uint32_t i = lane; i < 192; i += 64) base[i] = 0;
    ze_fence(); zh_sync();
    for (uint32_t n = lane; n <= maxSym; n += 64) zh_lds_atomic_inc(&base[ze_huf_bucket(count[n])]);
    ze_fence(); zh_sync();
    {   // suffix sums over the 192 buckets, three per lane
        const uint32_t a0 = base[3 * lane], a1 = base[3 * lane + 1], a2 = base[3 * lane + 2];
        const uint32_t incl = zh_scan_add(a0 + a1 + a2);
        const uint32_t above = zh_shfl(incl, 63) - incl;
        const uint32_t b2 = a2 + above, b1 = a1 + b2, b0 = a0 + b1;
        zh_sync();
        base[3 * lane] = b0; base[3 * lane + 1] = b1; base[3 * lane + 2] = b2;
        cur[3 * lane] = (uint16_t)b0; cur[3 * lane + 1] = (uint16_t)b1; cur[3 * lane + 2] = (uint16_t)b2;
        if (lane == 0) cur[192] = 0;
    }
    ze_fence(); zh_sync();
    for (uint32_t c0 = 0; c0 <= maxSym; c0 += 64) {                                // sorted position = symbols in higher buckets + same-bucket symbols below
        const uint32_t n = c0 + lane;
        const bool act = n <= maxSym;
        const uint32_t r = act ? ze_huf_bucket(count[n]) + 1 : 0u;
        const uint64_t same = ze_same_key(r, 8, act);
        uint32_t pos = 0;
        if (act) pos = (uint32_t)cur[r] + (uint32_t)zh_popc64(same & lt);
        zh_sync();
        if (act) {
            if ((same >> lane) >> 1 == 0) cur[r] = (uint16_t)(pos + 1);
            node[pos].count = count[n]; node[pos].byte = (uint8_t)n;
        }
        ze_fence(); zh_sync();
    }
    if (zh_opaque(lane) == 0) {
        for (uint32_t r = 166; r < 191; r++) {
            const int bsize = (int)cur[r] - (int)base[r];
            if (bsize > 1) ze_huf_quicksort(node + base[r], 0, bsize - 1, L.stack);
        }
        // the tree: leaves are taken from the small end of the sorted list, parents from the queue of nodes made so far (both in rising
        // count order); a tie takes the parent
        int last = (int)maxSym;
        while (node[last].count == 0) last--;
        int leaf = last, made = 256, root = made + leaf - 1, parent = made;
        node[made].count = node[leaf].count + node[leaf - 1].count;
        node[leaf].parent = node[leaf - 1].parent = (uint16_t)made;
        made++; leaf -= 2;
        for (int n = made; n <= root; n++) node[n].count = 1u << 30;
        node[-1].count = 1u << 31;                                                  // sentinel below leaf 0
        while (made <= root) {
            const int n1 = (node[leaf].count < node[parent].count) ? leaf-- : parent++;
            const int n2 = (node[leaf].count < node[parent].count) ? leaf-- : parent++;
            node[made].count = node[n1].count + node[n2].count;
            node[n1].parent = node[n2].parent = (uint16_t)made;
            made++;
        }
        L.misc[12] = (uint32_t)last; L.misc[13] = (uint32_t)root;
    }
    ze_fence(); zh_sync();
    const int last = (int)zh_first(L.misc[12]); const uint32_t root = zh_first(L.misc[13]);
    for (int n = (int)lane; n <= last; n += 64) {                                  // depth of every leaf
        uint32_t d = 0, q = (uint32_t)n;
        while (q != root) { q = node[q].parent; d++; }
        node[n].nbBits = (uint8_t)d;
    }
    ze_fence(); zh_sync();
    uint32_t largest = node[last].nbBits;
    zh_sync();                                                                      // (everyone has read it before lane 0 starts changing lengths)
    if (largest > maxBits) {
        if (zh_opaque(lane) == 0) {
            int totalCost = 0; const uint32_t baseCost = 1u << (largest - maxBits);
            int n = last;
            while (node[n].nbBits > maxBits) { totalCost += (int)(baseCost - (1u << (largest - node[n].nbBits))); node[n].nbBits = (uint8_t)maxBits; n--; }
            while (node[n].nbBits == maxBits) --n;
            totalCost >>= (largest - maxBits);
            const uint32_t none = 0xF0F0F0F0u; uint32_t* const rankLast = L.stack;      // 14 entries; the quicksort's stack is idle now
            for (int i = 0; i < 14; i++) rankLast[i] = none;
            {   uint32_t curBits = maxBits;
                for (int pos = n; pos >= 0; pos--) { if (node[pos].nbBits >= curBits) continue; curBits = node[pos].nbBits; rankLast[maxBits - curBits] = (uint32_t)pos; } }
            while (totalCost > 0) {
                uint32_t dec = (uint32_t)zh_highbit32((uint32_t)totalCost) + 1;
                for (; dec > 1; dec--) {
                    const uint32_t highPos = rankLast[dec], lowPos = rankLast[dec - 1];
                    if (highPos == none) continue;
                    if (lowPos == none) break;
                    if (node[highPos].count <= 2 * node[lowPos].count) break;
                }
                while (dec <= 12 && rankLast[dec] == none) dec++;
                totalCost -= 1 << (dec - 1);
                node[rankLast[dec]].nbBits++;
                if (rankLast[dec - 1] == none) rankLast[dec - 1] = rankLast[dec];
                if (rankLast[dec] == 0) rankLast[dec] = none;
                else { rankLast[dec]--; if (node[rankLast[dec]].nbBits != maxBits - dec) rankLast[dec] = none; }
            }
            while (totalCost < 0) {
                if (rankLast[1] == none) {
                    while (node[n].nbBits == maxBits) n--;
                    node[n + 1].nbBits--; rankLast[1] = (uint32_t)(n + 1); totalCost++;
                    continue;
                }
                node[rankLast[1] + 1].nbBits--; rankLast[1]++; totalCost++;
            }
        }
        largest = maxBits;
        ze_fence(); zh_sync();
    }
    // codes: per length, numbered in symbol order from a start value that comes from the counts of the longer lengths
    uint16_t* const perRank = (uint16_t*)(L.stack + 16); uint16_t* const start = perRank + 16;
    if (lane < 16) { perRank[lane] = 0; start[lane] = 0; }
    for (uint32_t i = lane; i < 256; i += 64) { L.hufBits[i] = 0; L.hufCode[i] = 0; }
    ze_fence(); zh_sync();
    {   uint32_t* const pr32 = (uint32_t*)(L.stack + 32);                          // 16 counters for the atomics
        if (lane < 16) pr32[lane] = 0;
        ze_fence(); zh_sync();
        for (int n = (int)lane; n <= last; n += 64) zh_lds_atomic_inc(&pr32[node[n].nbBits]);
        for (uint32_t n = lane; n <= maxSym; n += 64) L.hufBits[node[n].byte] = node[n].nbBits;
        ze_fence(); zh_sync();
        if (zh_opaque(lane) == 0) { uint16_t mn = 0; for (int r = (int)largest; r > 0; r--) { start[r] = mn; mn = (uint16_t)((mn + pr32[r]) >> 1); } }
        ze_fence(); zh_sync();
    }
    for (uint32_t c0 = 0; c0 <= maxSym; c0 += 64) {
        const uint32_t sy = c0 + lane;
        const bool act = sy <= maxSym;
        const uint32_t nb = act ? L.hufBits[sy] : 0u;
        const uint64_t same = ze_same_key(nb, 4, act);
        uint32_t code = 0;
        if (act && nb) code = (uint32_t)start[nb] + (uint32_t)zh_popc64(same & lt);
        zh_sync();
        if (act && nb) { if ((same >> lane) >> 1 == 0) start[nb] = (uint16_t)(code + 1); L.hufCode[sy] = (uint16_t)code; }
        ze_fence(); zh_sync();
    }
    return largest;
}

// The Huffman table description (HUF_writeCTable_wksp, zstd.c:17005): weights = log + 1 - code length, FSE-compressed when that is
// shorter (HUF_compressWeights :16904), else 4 bits each. All lanes call; 0 = cannot be described, 1 = one weight value only.
ZH_COLD uint32_t ze_huf_compress_weights(ZeLDS& L, uint8_t* out, const uint8_t* w, uint32_t n)
{
    const uint32_t lane = zh_lane();
    uint32_t* const count = &L.cnt[0][0];           // free here: the bucket sort is over, the sequence histograms come later
    if (n <= 1) return 0;
    zh_sync();
    if (lane < 13) count[lane] = 0;
    ze_fence(); zh_sync();
    for (uint32_t i = lane; i < n; i += 64) zh_lds_atomic_inc(&count[w[i]]);
    ze_fence(); zh_sync();
    const uint32_t c = lane <= 12 ? count[lane] : 0u;
    const uint64_t present = zh_ballot(c != 0);
    const uint32_t maxSym = present ? 63u - (uint32_t)zh_clz64(present) : 0u, maxCount = zh_wave_max(c);
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    const uint32_t lg = ze_fse_optimal_log(6, n, maxSym, 2);
    int16_t* const norm = ze_norm_area(L);
    if (ze_fse_normalize_wave(norm, lg, count, n, maxSym, false) < 0) return 0;
    ze_fence(); zh_sync();
    const uint32_t h = ze_fse_write_ncount_wave(out, norm, maxSym, lg, (uint32_t*)((uint8_t*)L.node + 2048));
    ZeCTab& t = L.tab[0];
    ze_fse_build_ctab_wave(t, norm, maxSym, lg, ze_cell_sym(L), (uint16_t*)L.stack, ze_fill_area(L));
    if (n <= 2) return 0;
    if (zh_opaque(lane) == 0) {                     // the weights themselves: two interleaved tANS states, last weight first (FSE_compress_usingCTable_generic, zstd.c:16488)
        ZeBits b; ze_bw_init(b, out + h, 512);
        uint32_t ip = n, s1, s2;
        if (n & 1) { s1 = ze_fse_first_state(t, w[--ip]); s2 = ze_fse_first_state(t, w[--ip]); s1 = ze_fse_encode(t, b, s1, w[--ip]); }
        else { s2 = ze_fse_first_state(t, w[--ip]); s1 = ze_fse_first_state(t, w[--ip]); }
        while (ip > 0) {
            s2 = ze_fse_encode(t, b, s2, w[--ip]);
            s1 = ze_fse_encode(t, b, s1, w[--ip]);
        }
        ze_bw_add(b, s2, lg); ze_bw_add(b, s1, lg);
        L.misc[12] = ze_bw_close(b);
    }
    ze_fence(); zh_sync();
    const uint32_t c2 = zh_first(L.misc[12]);
    zh_sync();
    return c2 ? h + c2 : 0;
}
ZH_DEVFN uint32_t ze_huf_write_table(ZeLDS& L, uint8_t* out, uint32_t maxSym, uint32_t lg)
{
    const uint32_t lane = zh_lane();
    uint8_t* w = ze_weights(L);
    zh_sync();
    for (uint32_t n = lane; n < maxSym; n += 64) w[n] = L.hufBits[n] ? (uint8_t)(lg + 1 - L.hufBits[n]) : (uint8_t)0;
    ze_fence(); zh_sync();
    const uint32_t h = ze_huf_compress_weights(L, out + 1, w, maxSym);
    if (h > 1 && h < maxSym / 2) { if (zh_opaque(lane) == 0) out[0] = (uint8_t)h; ze_fence(); zh_sync(); return h + 1; }
    if (maxSym > 128) return 0;
    if (zh_opaque(lane) == 0) { out[0] = (uint8_t)(128 + (maxSym - 1)); w[maxSym] = 0; }
    ze_fence(); zh_sync();
    for (uint32_t n = 2 * lane; n < maxSym; n += 128) out[n / 2 + 1] = (uint8_t)((w[n] << 4) + w[n + 1]);
    ze_fence(); zh_sync();
    return (maxSym + 1) / 2 + 1;
}

// LDS scratch that aliases the Huffman tree nodes (dead once the code lengths are final): first the literal code table
// (code | nbBits << 16 per byte value), later -- the literals being done -- the sequence encoder's scratch.
ZH_DEV uint32_t* ze_scratch(ZeLDS& L) { return (uint32_t*)L.node; }

// The four Huffman streams of a literals section (HUF_compress4X_usingCTable_internal zstd.c:17925; each stream is
// HUF_compress1X_usingCTable_internal_body :17813, last symbol first, closed by a 1 bit), encoded by the whole wave: 16 lanes
// per stream, each lane a contiguous run of the stream's symbols. A stream is a pure concatenation of codes, so a lane's bit
// offset is the code-length total of the runs behind it (wave prefix sum); lanes write whole dwords they own with plain stores
// and the two dwords they may share with a neighbour with atomic ORs into a zeroed area. ct = code | nbBits << 16 (LDS).
// All lanes call. Returns 6 + the four stream sizes, or 0 when a stream is too long for the jump table or the room.
ZH_COLD uint32_t ze_huf_encode_4x_wave(const uint32_t* ct, uint8_t* body, uint32_t bcap, const uint8_t* lit, uint32_t n)
{
    const uint32_t lane = zh_lane();
    const uint32_t strm = lane >> 4, sub = lane & 15;
    const uint32_t seg = (n + 3) / 4;
    const uint32_t s0 = strm * seg, len = strm < 3 ? seg : n - 3 * seg;
    const uint32_t run = (len + 15) / 16;
    const uint32_t ra = sub * run < len ? sub * run : len, rb = ra + run < len ? ra + run : len;
    const uint8_t* p = lit + s0;
    uint32_t bits = 0;
    {   uint32_t i = rb;
        while (i - ra >= 4) { const uint32_t w = zh_ld32(p + i - 4); i -= 4;
            bits += (ct[w >> 24] >> 16) + (ct[(w >> 16) & 255] >> 16) + (ct[(w >> 8) & 255] >> 16) + (ct[w & 255] >> 16); }
        while (i > ra) { --i; bits += ct[p[i]] >> 16; } }
    const uint32_t incl = zh_scan_add(bits);
    const uint32_t gEnd = zh_shfl(incl, strm * 16 + 15);
    const uint32_t gBefore = zh_shfl(incl, strm ? strm * 16 - 1 : 0);
    const uint32_t T = gEnd - (strm ? gBefore : 0u);                 // bits of my stream
    const uint32_t off = gEnd - incl;                                // bits emitted before mine: the runs behind me
    const uint32_t mySize = (T + 1 + 7) / 8;
    const uint32_t z1 = zh_shfl(mySize, 0), z2 = zh_shfl(mySize, 16), z3 = zh_shfl(mySize, 32), z4 = zh_shfl(mySize, 48);
    if (z1 > 65535 || z2 > 65535 || z3 > 65535 || z4 > 65535 || n < 12 || 6 + z1 + z2 + z3 + z4 > bcap) return 0;
    const uint32_t total = 6 + z1 + z2 + z3 + z4;
    {   // zero the stream area byte-exactly (neighbouring bytes belong to headers already written / to other frames)
        uint8_t* z0 = body + 6; uint8_t* const ze = body + total;
        const uint32_t head = (uint32_t)((4 - ((uintptr_t)z0 & 3)) & 3);
        const uint32_t h = head < (uint32_t)(ze - z0) ? head : (uint32_t)(ze - z0);
        if (lane < h) z0[lane] = 0;
        z0 += h;
        const uint32_t nd = (uint32_t)(ze - z0) >> 2;
        for (uint32_t i = lane; i < nd; i += 64) ((uint32_t*)z0)[i] = 0;
        z0 += 4 * (size_t)nd;
        if (lane < (uint32_t)(ze - z0)) z0[lane] = 0;
    }
    ze_fence();
    zh_sync();
    {
        const uint32_t so = 6 + (strm > 0 ? z1 : 0u) + (strm > 1 ? z2 : 0u) + (strm > 2 ? z3 : 0u);
        const uintptr_t A = (uintptr_t)(body + so);
        const uint32_t sh0 = (uint32_t)(A & 3) * 8 + off;            // bit offset from the aligned dword at or below the stream start
        uint32_t* d = (uint32_t*)(A & ~(uintptr_t)3) + (sh0 >> 5);
        uint32_t nacc = sh0 & 31; uint64_t acc = 0; bool first = true;
#define ZE_HEMIT() do { if (first) { zh_atomic_or(d, (uint32_t)acc); first = false; } else *d = (uint32_t)acc; d++; acc >>= 32; nacc -= 32; } while (0)
#define ZE_HSYM(s) do { const uint32_t e_ = ct[(s)]; acc |= (uint64_t)(e_ & 0xFFFFu) << nacc; nacc += e_ >> 16; if (nacc >= 32) ZE_HEMIT(); } while (0)
        uint32_t i = rb;
        while (i - ra >= 4) { const uint32_t w = zh_ld32(p + i - 4); i -= 4;
            ZE_HSYM(w >> 24); ZE_HSYM((w >> 16) & 255); ZE_HSYM((w >> 8) & 255); ZE_HSYM(w & 255); }
        while (i > ra) { --i; ZE_HSYM(p[i]); }
        if (sub == 0) { acc |= 1ull << nacc; nacc++; if (nacc >= 32) ZE_HEMIT(); }     // end mark after the stream's first symbol
        if (acc) zh_atomic_or(d, (uint32_t)acc);
#undef ZE_HSYM
#undef ZE_HEMIT
        if (lane < 3) zh_st16(body + 2 * lane, (uint16_t)(lane == 0 ? z1 : lane == 1 ? z2 : z3));
    }
    ze_fence();
    zh_sync();
    return total;
}

// One Huffman stream (HUF_compress1X_usingCTable_internal_body, zstd.c:17813: last symbol first, closed by a 1 bit) by the whole wave --
// the single-stream form of the above: 64 lanes, each a contiguous run; no jump table. The
// lane-0 encoder reads the literals a byte at a time from global memory -- a dependent round trip per literal -- which was 40 % of
// the entropy kernel on small inputs with a dictionary (sections of fewer than 1 024 literals are single streams; r02d). Returns
// the stream size, 0 when it does not fit.
ZH_COLD uint32_t ze_huf_encode_1x_wave(const uint32_t* ct, uint8_t* body, uint32_t bcap, const uint8_t* lit, uint32_t n)
{
    const uint32_t lane = zh_lane();
    const uint32_t run = (n + 63) / 64;
    const uint32_t ra = lane * run < n ? lane * run : n, rb = ra + run < n ? ra + run : n;
    uint32_t bits = 0;
    for (uint32_t i = ra; i < rb; i++) bits += ct[lit[i]] >> 16;
    const uint32_t incl = zh_scan_add(bits);
    const uint32_t T = zh_shfl(incl, 63);
    const uint32_t off = T - incl;                                   // bits emitted before mine: the runs behind me
    const uint32_t total = (T + 1 + 7) / 8;
    if (total > bcap) return 0;
    {   // zero the stream area byte-exactly
        uint8_t* z0 = body; uint8_t* const ze = body + total;
        const uint32_t head = (uint32_t)((4 - ((uintptr_t)z0 & 3)) & 3);
        const uint32_t h = head < (uint32_t)(ze - z0) ? head : (uint32_t)(ze - z0);
        if (lane < h) z0[lane] = 0;
        z0 += h;
        const uint32_t nd = (uint32_t)(ze - z0) >> 2;
        for (uint32_t i = lane; i < nd; i += 64) ((uint32_t*)z0)[i] = 0;
        z0 += 4 * (size_t)nd;
        if (lane < (uint32_t)(ze - z0)) z0[lane] = 0;
    }
    ze_fence();
    zh_sync();
    {
        const uintptr_t A = (uintptr_t)body;
        const uint32_t sh0 = (uint32_t)(A & 3) * 8 + off;
        uint32_t* d = (uint32_t*)(A & ~(uintptr_t)3) + (sh0 >> 5);
        uint32_t nacc = sh0 & 31; uint64_t acc = 0; bool first = true;
        for (uint32_t i = rb; i > ra;) {
            --i;
            const uint32_t e = ct[lit[i]];
            acc |= (uint64_t)(e & 0xFFFFu) << nacc; nacc += e >> 16;
            if (nacc >= 32) { if (first) { zh_atomic_or(d, (uint32_t)acc); first = false; } else *d = (uint32_t)acc; d++; acc >>= 32; nacc -= 32; }
        }
        if (lane == 0) {                                            // end mark after the stream's first symbol
            acc |= 1ull << nacc; nacc++;
            if (nacc >= 32) { if (first) { zh_atomic_or(d, (uint32_t)acc); first = false; } else *d = (uint32_t)acc; d++; acc >>= 32; nacc -= 32; }
        }
        if (acc) zh_atomic_or(d, (uint32_t)acc);
    }
    ze_fence();
    zh_sync();
    return total;
}

// n bytes from s to d by the whole wave: 16 bytes per lane per step, four steps' loads in flight before the first store (a byte
// per lane per step is one memory round trip per 64 bytes: 2 048 of them for a 128 KiB block). Ranges must not overlap.
ZH_DEV void ze_copy_wave(uint8_t* d, const uint8_t* s, uint32_t n)
{
    const uint32_t lane = zh_lane();
    const uint32_t ng = n >> 4;
    for (uint32_t g = lane; g < ng; g += 256) {
        zh_v16 v0 = zh_ld128(s + 16 * (size_t)g), v1 = v0, v2 = v0, v3 = v0;
        if (g + 64 < ng) v1 = zh_ld128(s + 16 * (size_t)(g + 64));
        if (g + 128 < ng) v2 = zh_ld128(s + 16 * (size_t)(g + 128));
        if (g + 192 < ng) v3 = zh_ld128(s + 16 * (size_t)(g + 192));
        *(zh_v16*)(d + 16 * (size_t)g) = v0;
        if (g + 64 < ng) *(zh_v16*)(d + 16 * (size_t)(g + 64)) = v1;
        if (g + 128 < ng) *(zh_v16*)(d + 16 * (size_t)(g + 128)) = v2;
        if (g + 192 < ng) *(zh_v16*)(d + 16 * (size_t)(g + 192)) = v3;
    }
    const uint32_t done = ng << 4;
    if (lane < n - done) d[done + lane] = s[done + lane];
}

// literals header for raw / rle sections (ZSTD_noCompressLiterals :20842, ZSTD_compressRleLiteralsBlock :20884). All lanes call;
// returns the section size.
ZH_DEV uint32_t ze_plain_literals(uint8_t* out, const uint8_t* lit, uint32_t n, uint32_t type, bool rle)
{
    const uint32_t fl = 1 + (n > 31) + (n > 4095);
    if (zh_opaque(zh_lane()) == 0) {
        if (fl == 1) out[0] = (uint8_t)(type + (n << 3));
        else if (fl == 2) zh_st16(out, (uint16_t)(type + (1 << 2) + (n << 4)));
        else zh_st32(out, type + (3u << 2) + (n << 4));
        if (rle) out[fl] = lit[0];
    }
    if (rle) return fl + 1;
    ze_copy_wave(out + fl, lit, n);
    return fl + n;
}

// Byte histogram of p[0 .. n) into hist (LDS, 256 counters, zeroed by the caller; HIST_count_wksp zstd.c:16714). All lanes call.
// Sixteen bytes per lane per round with the next round's load already in flight: one byte per lane per round (the obvious loop)
// spends a global-memory round trip per 64 bytes and was 40 % of the entropy kernel.
ZH_DEV void ze_byte_hist(uint32_t* hist, const uint8_t* p, uint32_t n)
{
    const uint32_t lane = zh_lane();
    const uint32_t ng = n >> 4;
    uint32_t g = lane;
    zh_v16 cur; cur.lo = 0; cur.hi = 0;
    if (g < ng) cur = zh_ld128(p + 16 * (size_t)g);
    while (g < ng) {
        const uint32_t g2 = g + 64;
        zh_v16 nxt; nxt.lo = 0; nxt.hi = 0;
        if (g2 < ng) nxt = zh_ld128(p + 16 * (size_t)g2);
        uint64_t w = cur.lo;
        for (int k = 0; k < 8; k++) { zh_lds_atomic_inc(&hist[(uint32_t)w & 255]); w >>= 8; }
        w = cur.hi;
        for (int k = 0; k < 8; k++) { zh_lds_atomic_inc(&hist[(uint32_t)w & 255]); w >>= 8; }
        cur = nxt; g = g2;
    }
    const uint32_t done = ng << 4;
    if (lane < n - done) zh_lds_atomic_inc(&hist[p[done + lane]]);
}

// ZSTD_compressLiterals (zstd.c:20932) + HUF_compress_internal (:18089). The only "previous" Huffman table a single-block
// frame can have is the dictionary's (cd, may be null): repeat 0 none, 1 usable after validation, 2 valid.
// All lanes call; returns the literals-section size (uniform).
ZH_DEVFN uint32_t ze_compress_literals(ZeLDS& L, uint8_t* out, uint32_t cap, const uint8_t* lit, uint32_t n, uint32_t nbSeq, const ZePrevHuf* cd,
                                       uint32_t* pNewMaxSym /* 0xFFFFFFFF unless the section carries a freshly built table */, ZeProf* P = nullptr)
{
    *pNewMaxSym = 0xFFFFFFFFu;
    const uint32_t lane = zh_lane();
    const uint32_t lh = 3 + (n >= 1024) + (n >= 16384);
    uint32_t repeat = cd ? cd->repeat : 0;
    bool single = n < 256;
    if (repeat == 2 && lh == 3) single = true;
    const bool preferRepeat = n <= 1024;
    const uint32_t minLits = repeat == 2 ? 6 : 64;
    // decision: 0 raw, 1 rle, 2 try Huffman
    uint32_t decision = 2;
    bool useOld = false;
    if (n < minLits) decision = 0;
    const bool suspect = (nbSeq == 0) || (n / nbSeq >= 20);
    zh_sync();
    for (uint32_t i = lane; i < 256; i += 64) L.hist[i] = 0;
    zh_sync();
    uint32_t h = 0, builtMaxSym = 0;
    if (decision == 2 && preferRepeat && repeat == 2) useOld = true;          // no statistics needed
    else if (decision == 2) {
        if (suspect && n >= 4096 * 10) {
            // two 4 KiB samples (HUF_flags_suspectUncompressible)
            uint32_t total = 0;
            for (int part = 0; part < 2; part++) {
                const uint8_t* p = part ? lit + n - 4096 : lit;
                ze_byte_hist(L.hist, p, 4096);
                zh_sync();
                uint32_t m = 0;
                for (uint32_t i = lane; i < 256; i += 64) { if (L.hist[i] > m) m = L.hist[i]; }
                m = zh_wave_max(m);
                total += m;
                zh_sync();
                for (uint32_t i = lane; i < 256; i += 64) L.hist[i] = 0;
                zh_sync();
            }
            if (total <= ((2 * 4096) >> 7) + 4) decision = 0;
        }
        uint32_t maxSym = 0, largest = 0;
        if (decision == 2) {
            ze_byte_hist(L.hist, lit, n);
            zh_sync();
            uint32_t m = 0, ms = 0;
            for (uint32_t i = lane; i < 256; i += 64) { const uint32_t c = L.hist[i]; if (c > m) m = c; if (c) ms = i; }
            largest = zh_wave_max(m); maxSym = zh_wave_max(ms);
            if (largest == n) decision = 1;
            else if (largest <= (n >> 7) + 4) decision = 0;
        }
        zh_sync();
        if (decision == 2 && repeat == 1) {
            // HUF_validateCTable (zstd.c:17693): every present symbol needs a code in the old table
            uint32_t bad = cd->maxSym < maxSym ? 1u : 0u;
            for (uint32_t i = lane; i <= maxSym; i += 64) if (L.hist[i] && !cd->bits[i]) bad = 1;
            if (zh_wave_max(bad)) repeat = 0;
        }
        if (decision == 2 && preferRepeat && repeat != 0) useOld = true;
        else if (decision == 2) {
            uint32_t lg = 0;
            ZE_T(P, ZEP_LITSTAT);
            lg = ze_fse_optimal_log(11, n, maxSym, 1);
            lg = ze_huf_build(L, maxSym, lg);
            h = ze_huf_write_table(L, out + lh, maxSym, lg);
            zh_sync();
            ZE_T(P, ZEP_HUFBUILD);
            builtMaxSym = maxSym;
            if (h == 0) decision = 0;
            else if (repeat != 0) {
                // HUF_estimateCompressedSize (zstd.c:17681) of both tables on this histogram
                uint32_t ob = 0, nb = 0;
                for (uint32_t i = lane; i <= maxSym; i += 64) { ob += (uint32_t)cd->bits[i] * L.hist[i]; nb += (uint32_t)L.hufBits[i] * L.hist[i]; }
                ob = zh_scan_add(ob); nb = zh_scan_add(nb);
                const uint32_t oldSize = zh_shfl(ob, 63) >> 3, newSize = zh_shfl(nb, 63) >> 3;
                if (oldSize <= h + newSize || h + 12 >= n) useOld = true;
            }
            if (decision == 2 && !useOld) {
                if (h + 12 >= n) decision = 0;
                else repeat = 0;
            }
        }
    }
    if (decision == 2) {
        if (useOld) {
            zh_sync();
            for (uint32_t i = lane; i < 256; i += 64) { L.hufBits[i] = cd->bits[i]; L.hufCode[i] = cd->code[i]; }
            zh_sync();
            h = 0;
        }
        uint8_t* body = out + lh + h;
        const uint32_t bcap = cap - lh - h;
        uint32_t total = 0;
        {
            // one or four streams, wave-parallel
            uint32_t* ct = ze_scratch(L);
            zh_sync();
            for (uint32_t i = lane; i < 256; i += 64) ct[i] = (uint32_t)L.hufCode[i] | ((uint32_t)L.hufBits[i] << 16);
            zh_sync();
            total = single ? ze_huf_encode_1x_wave(ct, body, bcap, lit, n) : ze_huf_encode_4x_wave(ct, body, bcap, lit, n);
            ZE_T(P, ZEP_HUFENC);
        }
        uint32_t cl = total ? h + total : 0;
        if (cl >= n - 1) cl = 0;
        if (cl == 0 || cl >= n - ((n >> 6) + 2)) decision = 0;
        else {
            const uint32_t hType = repeat != 0 ? 3u : 2u;
            if (hType == 2) *pNewMaxSym = builtMaxSym;
            if (zh_opaque(lane) == 0) {
                if (lh == 3) { const uint32_t v = hType + ((uint32_t)(!single) << 2) + (n << 4) + (cl << 14); out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16); }
                else if (lh == 4) zh_st32(out, hType + (2u << 2) + (n << 4) + (cl << 18));
                else { zh_st32(out, hType + (3u << 2) + (n << 4) + (cl << 22)); out[4] = (uint8_t)(cl >> 10); }
            }
            zh_sync();
            return lh + cl;
        }
    }
    // raw or rle literals
    zh_sync();
    const uint32_t r = ze_plain_literals(out, lit, n, decision == 1 ? 1u : 0u, decision == 1);
    ze_fence();
    zh_sync();
    return r;
}

// ------------------------------------------------------------------------------------------ double-fast match finder (lane 0)
ZH_DEV uint32_t ze_hash(const uint8_t* p, int hbits, int mls)      // ZSTD_hashPtr, zstd.c:20084
{
    const uint64_t u = zh_ld64(p);
    switch (mls) {
    case 5: return (uint32_t)(((u << 24) * 889523592379ull) >> (64 - hbits));
    case 6: return (uint32_t)(((u << 16) * 227718039650203ull) >> (64 - hbits));
    case 7: return (uint32_t)(((u << 8) * 58295818150454627ull) >> (64 - hbits));
    case 8: return (uint32_t)((u * 0xCF1BBCDCB7A56463ull) >> (64 - hbits));
    default: return (uint32_t)(((uint32_t)u * 2654435761u) >> (32 - hbits));
    }
}
ZH_DEV uint32_t ze_common_len(const uint8_t* a, const uint8_t* b, const uint8_t* aend)     // ZSTD_count, zstd.c:20008
{
    const uint8_t* s = a;
    while (a + 8 <= aend) {
        const uint64_t d = zh_ld64(a) ^ zh_ld64(b);
        if (d) return (uint32_t)(a - s) + (uint32_t)(zh_ctz64(d) >> 3);
        a += 8; b += 8;
    }
    while (a < aend && *a == *b) { a++; b++; }
    return (uint32_t)(a - s);
}

// ZSTD_compressBlock_doubleFast_noDict_generic (zstd.c:31039) for a block that is the whole frame. Table cells hold
// position + 2 (0 = empty), so the reference's index comparisons keep their meaning with lowest == 2. lane 0 only.
// seqs: packed with ZE_SEQ_PACK. Returns nbSeq; *pLit = literal count.
// `frame` = first byte of the frame (index 2), `src` = the block; rep[] = the two repcodes in and out (offsets larger than the
// history are parked and restored, zstd.c:31091-31098, :31175-31182). Tables are the caller's: zeroed before a frame's first block.
ZH_DEVFN uint32_t ze_dfast_g(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* frame, const uint8_t* src, uint32_t srcSize, const ZePar& cp,
                             uint32_t* hashLong, uint32_t* hashSmall, uint32_t* rep)
{
    const int hl = cp.hlog, hs = cp.clog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint8_t* const base = frame - 2;
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    // lowest index a match may start at: the frame start, or what the window still covers at the END of this block
    // (ZSTD_getLowestPrefixIndex zstd.c:20566 after ZSTD_window_enforceMaxDist :20386)
    const uint32_t maxDist = 1u << cp.wlog;
    const uint32_t endIndex = (uint32_t)(iend - base);
    const uint32_t LOW = endIndex - 2 > maxDist ? endIndex - maxDist : 2;
    const uint8_t* anchor = src;
    // the first position is skipped when the block starts with an EMPTY prefix (zstd.c:31091 `ip += (dictAndPrefixLength == 0)`): the frame's
    // first block -- and a later block whose window reaches back exactly to its own start (window == block size, explicit window_log 17)
    const uint8_t* ip = src + ((uint32_t)(src - base) == LOW ? 1 : 0);
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    {   const uint32_t c0 = (uint32_t)(ip - base); const uint32_t maxRep = c0 - 2 > maxDist ? maxDist : c0 - 2;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; } }
    uint32_t nseq = 0; uint8_t* lp = lits;
    if (srcSize < 8) { for (uint32_t i = 0; i < srcSize; i++) lp[i] = src[i]; *pLit = srcSize; return 0; }
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = anchor[i_]; lp += ll_; \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
    for (;;) {
        uint32_t step = 1; const uint8_t* nextStep = ip + 256; const uint8_t* ip1 = ip + step;
        uint32_t mLength = 0, offset = 0, curr = 0;
        if (ip1 > ilimit) break;
        uint32_t hl0 = ze_hash(ip, hl, 8), idxl0 = hashLong[hl0];
        uint32_t hl1 = 0, idxl1 = 0;
        int found = 0;
        do {
            const uint32_t hs0 = ze_hash(ip, hs, mls), idxs0 = hashSmall[hs0];
            curr = (uint32_t)(ip - base);
            hashLong[hl0] = curr; hashSmall[hs0] = curr;
            if (off1 > 0 && zh_ld32(ip + 1 - off1) == zh_ld32(ip + 1)) {
                mLength = ze_common_len(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                ZE_STORE(ip - anchor, 1, mLength);
                found = 1; break;
            }
            hl1 = ze_hash(ip1, hl, 8);
            if (idxl0 >= LOW && zh_ld64(base + idxl0) == zh_ld64(ip)) {
                const uint8_t* m = base + idxl0;
                mLength = ze_common_len(ip + 8, m + 8, iend) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > base + LOW && ip[-1] == m[-1]) { ip--; m--; mLength++; }
                found = 2; break;
            }
            idxl1 = hashLong[hl1];
            if (idxs0 >= LOW && zh_ld32(base + idxs0) == zh_ld32(ip)) {
                const uint8_t* m = base + idxs0;
                mLength = ze_common_len(ip + 4, m + 4, iend) + 4;
                offset = (uint32_t)(ip - m);
                if (idxl1 > LOW && zh_ld64(base + idxl1) == zh_ld64(ip1)) {
                    const uint8_t* m1 = base + idxl1;
                    const uint32_t l1 = ze_common_len(ip1 + 8, m1 + 8, iend) + 8;
                    if (l1 > mLength) { ip = ip1; mLength = l1; offset = (uint32_t)(ip - m1); m = m1; }
                }
                while (ip > anchor && m > base + LOW && ip[-1] == m[-1]) { ip--; m--; mLength++; }
                found = 2; break;
            }
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
        } while (ip1 <= ilimit);
        if (!found) break;
        if (found == 2) {
            off2 = off1; off1 = offset;
            if (step < 4) hashLong[hl1] = (uint32_t)(ip1 - base);
            ZE_STORE(ip - anchor, offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = curr + 2;
            hashLong[ze_hash(base + ins, hl, 8)] = ins;
            hashLong[ze_hash(ip - 2, hl, 8)] = (uint32_t)(ip - 2 - base);
            hashSmall[ze_hash(base + ins, hs, mls)] = ins;
            hashSmall[ze_hash(ip - 1, hs, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit && off2 > 0 && zh_ld32(ip) == zh_ld32(ip - off2)) {
                const uint32_t r = ze_common_len(ip + 4, ip + 4 - off2, iend) + 4;
                const uint32_t t = off2; off2 = off1; off1 = t;
                hashSmall[ze_hash(ip, hs, mls)] = (uint32_t)(ip - base);
                hashLong[ze_hash(ip, hl, 8)] = (uint32_t)(ip - base);
                ZE_STORE(0, 1, r);
                ip += r; anchor = ip;
            }
        }
    }
#undef ZE_STORE
    {   const uint32_t lastLL = (uint32_t)(iend - anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = anchor[i]; lp += lastLL; }
    *pLit = (uint32_t)(lp - lits);
    if (saved1 != 0 && off1 != 0) saved2 = saved1;
    rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
    return nseq;
}
ZH_DEV uint32_t ze_dfast(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* src, uint32_t srcSize, const ZePar& cp,
                         uint32_t* hashLong, uint32_t* hashSmall)
{
    uint32_t rep[2] = {1, 4};
    return ze_dfast_g(seqs, lits, pLit, src, src, srcSize, cp, hashLong, hashSmall, rep);
}


// ------------------------------------------------------------------------------------------ E1, flat form (double-fast, no dictionary)
// ZSTD_compressBlock_doubleFast_noDict_generic (zstd.c:31039) for a frame of one block, one LANE per frame and every lane of the
// wave busy: the reference's nested loops (probe until a match, then extend, store, insert, repeat-offset loop) are flattened into
// ONE loop whose every trip is a probe step that may end in a match, so lanes never wait for their neighbours to find a match.
// Each trip is two dependent memory rounds -- the table cells, then the candidates' bytes -- and, on a match, the extension /
// insertion reads, which hit the lines just fetched. The long-hash candidate of ip+step is fetched once and carried into the next
// trip, where it is the long-hash candidate of ip (zstd.c:31208-31211 carries hl1 / idxl1 the same way). Literals are not copied:
// the entropy kernel gathers them from the sequence list. Positions are frame-relative; cells hold position + 2 (0 = empty), so
// the reference's comparisons against the lowest prefix index (2) keep their meaning. srcSize in [16, 128 KiB]; tables zeroed.
#ifdef ZHIP_EMU
#define ZE_STAT(i) (zd_stat[i]++)
#else
#define ZE_STAT(i) ((void)0)
#endif
ZH_DEV uint32_t ze_hl8(uint64_t u, uint32_t sh) { return (uint32_t)((u * 0xCF1BBCDCB7A56463ull) >> 32) >> sh; }
ZH_DEV uint32_t ze_hsx(uint64_t u, uint32_t shl, uint64_t prime, uint32_t sh) { return (uint32_t)(((u << shl) * prime) >> 32) >> sh; }
ZH_DEV uint32_t ze_count_fwd(const uint8_t* src, uint32_t a, uint32_t b, uint32_t srcSize)     // ZSTD_count (zstd.c:20008): b < a, a + result <= srcSize
{
    uint32_t len = 0;
    while (a + len + 32 <= srcSize) {                                  // 32 bytes a round: most matches end inside the first
        ZE_STAT(12);
        const zh_v16 x0 = zh_ld128(src + a + len), y0 = zh_ld128(src + b + len), x1 = zh_ld128(src + a + len + 16), y1 = zh_ld128(src + b + len + 16);
        const uint64_t d0 = x0.lo ^ y0.lo, d1 = x0.hi ^ y0.hi, d2 = x1.lo ^ y1.lo, d3 = x1.hi ^ y1.hi;
        if (d0) return len + (uint32_t)(zh_ctz64(d0) >> 3);
        if (d1) return len + 8 + (uint32_t)(zh_ctz64(d1) >> 3);
        if (d2) return len + 16 + (uint32_t)(zh_ctz64(d2) >> 3);
        if (d3) return len + 24 + (uint32_t)(zh_ctz64(d3) >> 3);
        len += 32;
    }
    while (a + len + 8 <= srcSize) {
        const uint64_t d = zh_ld64(src + a + len) ^ zh_ld64(src + b + len);
        if (d) return len + (uint32_t)(zh_ctz64(d) >> 3);
        len += 8;
    }
    while (a + len < srcSize && src[a + len] == src[b + len]) len++;
    return len;
}
// Cells of the flat kernel's tables: position + 2 in the low 18 bits (one block: <= 131 074), and in the 14 bits above a TAG of
// the bytes the cell's owner hashed -- more bits of the same product for the long table (so equal 8 bytes give equal tags), a hash
// of the first 4 bytes for the short table (the short check compares 4 bytes, zstd.c:31167). A candidate whose tag differs cannot
// pass the reference's byte comparison, so its bytes are never fetched: the decisions are the reference's, most failed probes
// cost no candidate read (libzstd does the same for dictionary tables, ZSTD_SHORT_CACHE_TAG_BITS zstd.c:20636).
#define ZE_CELL_IDX(c) ((c) & 0x3FFFFu)
// PB = the position's bits in a cell (18: sources of one block; 22: frames of several blocks up to 4 MiB - 2, the tag then has 10 bits).
// BLK = true searches ONE BLOCK [bs, be) of a larger frame -- tables as the blocks before left them, `rep` = the repeat offsets in and out
// (ZSTD_compressBlock_doubleFast_noDict_generic's entry and _cleanup, zstd.c:31091-31098 / :31252-31258); false: a whole source [0, be).
// Every trip examines NP consecutive probe positions in the same three memory rounds (own bytes -> table cells -> plausible candidates' bytes): four
// probes in five fail, and after a failed probe the next one is fully determined (its position, its step, and -- through the forwarding selects below --
// the table state it must see), so the probes after the first are speculative and simply dropped when an earlier one matches. The order of table reads
// and writes is the reference's: probe k's cell reads see the writes of the probes before it in the trip, probe k's writes happen only if every probe
// before it failed. NP = 2 consumes 1.8 probes per trip on the bench corpus, 3: 2.44, 4: 2.95 -- for 1.5 x / 2 x the loads, most of them dropped: batches bound
// by ONE source's serial chain (up to ~32 768 sources per launch) take four, launches of up to 65 536 three, larger ones -- bound by transactions -- two
// (r04zd / r04zg). The function was written for two probes in round 1 and generalised in round 4; at NP = 2 it compiles to the same kernel (4 573 against
// 4 569 instructions, 85 against 84 VGPRs).
template <int PB, bool BLK, int NP>
ZH_DEV uint32_t ze_dfast_flat_np(uint64_t* seqs, const uint8_t* src, uint32_t bs, uint32_t be, int hlog, int clog, int mml, uint32_t* hashLong, uint32_t* hashSmall, uint32_t* rep, const uint8_t* idle, uint32_t epoch = 0)
{
    // (round 6) sources of one block: the top EB = 6 of the 14 bits above the position hold the LAUNCH's number (ZhipEncodeArgs.tabEpoch, 0 .. 63), the 8 below them the tag. A cell
    // another launch wrote fails the tag comparison like any cell of other bytes -- "cannot match", what a zeroed cell says too -- so the tables need no zeroing between launches
    // (the host zeroes an allocation once and every 63 launches); an 8-bit tag still keeps 255 of 256 foreign candidates from being fetched. Number 0 = tables zeroed per launch, as before.
    constexpr uint32_t TBA = 32 - PB, EB = PB == 18 ? 6u : 0u, TB = TBA - EB, TM = (1u << TB) - 1;
    const uint32_t EPW = EB ? (epoch & ((1u << EB) - 1u)) << (PB + TB) : 0u;
#undef ZE_CELL_IDX
#define ZE_CELL_IDX(c) ((c) & ((1u << PB) - 1))
    const uint32_t shL = 32u - (uint32_t)hlog, shS = 32u - (uint32_t)clog;
    const int mls = mml <= 4 ? 4 : mml >= 7 ? 7 : mml;
    const uint32_t shlS = mls == 4 ? 32u : (uint32_t)(64 - 8 * mls);
    const uint64_t primeS = mls == 4 ? 2654435761ull : mls == 5 ? 889523592379ull : mls == 6 ? 227718039650203ull : 58295818150454627ull;
#define ZE_PL(u) ((uint32_t)(((u) * 0xCF1BBCDCB7A56463ull) >> 32))
#define ZE_PS(u) ((uint32_t)((((u) << shlS) * primeS) >> 32))
#define ZE_TL(ph) (((((ph) >> (shL - TB)) & TM) << PB) | EPW)
#define ZE_TS(u) ((((((uint32_t)(u) * 2654435761u) >> (shS - TB)) & TM) << PB) | EPW)
    const uint32_t ilimit = be - 8, srcSize = be;
    uint32_t ip = BLK ? bs + (bs == 0 ? 1u : 0u) : 1u, anchor = BLK ? bs : 0u, off1 = 1, off2 = 0, nseq = 0;
    uint32_t saved1 = 0, saved2 = 0;
    if (BLK) {                                                        // as ze_dfast_flat_t: one block [bs, be) of a larger frame, repeat offsets in and out
        off1 = rep[0]; off2 = rep[1];
        if (off2 > ip) { saved2 = off2; off2 = 0; }
        if (off1 > ip) { saved1 = off1; off1 = 0; }
        if (be < bs + 8) { return 0; }
    }
    uint32_t step = 1, nextStep = 0, cellL0 = 0, pl0 = 0; uint64_t cl0 = 0;
    bool fresh = true;
    for (;;) {
        if (fresh) { step = 1; nextStep = ip + 256; }
        // the trip's positions: probe k at pos[k], with the step a failed probe k - 1 leaves behind (zstd.c:31207); pos[NP] is where the search goes on
        uint32_t pos[NP + 1], st[NP + 1], nx[NP + 1];
        pos[0] = ip; st[0] = step; nx[0] = nextStep;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            pos[k + 1] = pos[k] + st[k]; st[k + 1] = st[k]; nx[k + 1] = nx[k];
            if (pos[k + 1] >= nx[k]) { st[k + 1]++; nx[k + 1] += 256; }
        }
        if (pos[1] > ilimit) break;
        // R = the probes of this trip that are real if all before them fail: probe k needs pos[k + 1] <= ilimit (the reference's loop condition)
        uint32_t R = 1;
#pragma unroll
        for (int k = 1; k < NP; k++) if (R == (uint32_t)k && pos[k + 1] <= ilimit) R = (uint32_t)k + 1;
        ZE_STAT(10);
        // round 0: the probes' own bytes (positions beyond the last real one read the last valid position: ignored)
        uint64_t w[NP + 1]; uint32_t rp[NP], pL[NP + 1], hl[NP + 1], hs[NP];
#pragma unroll
        for (int k = 0; k <= NP; k++) { const uint32_t q = pos[k] <= ilimit ? pos[k] : ilimit; w[k] = zh_ld64(src + q); }
#pragma unroll
        for (int k = 0; k < NP; k++) { const uint32_t q = pos[k] <= ilimit ? pos[k] : ilimit; rp[k] = zh_ld32(src + q + 1 - off1); }
#pragma unroll
        for (int k = 0; k <= NP; k++) { pL[k] = ZE_PL(w[k]); hl[k] = pL[k] >> shL; }
#pragma unroll
        for (int k = 0; k < NP; k++) hs[k] = ZE_PS(w[k]) >> shS;
        // round 1: the table cells, all in flight together. The long cell of probe 0 was read one trip earlier unless the trip is fresh.
        uint32_t cL[NP + 1], cS[NP], newL[NP + 1], newS[NP];
        const uint32_t tA = hashLong[fresh ? hl[0] : hl[1]];
#pragma unroll
        for (int k = 0; k < NP; k++) cS[k] = hashSmall[hs[k]];
#pragma unroll
        for (int k = 1; k <= NP; k++) cL[k] = hashLong[hl[k]];
        if (fresh) cellL0 = tA;
        cL[0] = cellL0;
#pragma unroll
        for (int k = 0; k <= NP; k++) newL[k] = (pos[k] + 2) | ZE_TL(pL[k]);
#pragma unroll
        for (int k = 0; k < NP; k++) newS[k] = (pos[k] + 2) | ZE_TS(w[k]);
        // what the reference's later reads see after its earlier writes of this trip (zstd.c:31121 then :31163): the latest earlier probe with the same cell wins
#pragma unroll
        for (int k = 1; k <= NP; k++) {
#pragma unroll
            for (int j = 0; j < k; j++) {
                if (hl[k] == hl[j]) cL[k] = newL[j];
                if (k < NP && hs[k < NP ? k : 0] == hs[j]) cS[k < NP ? k : 0] = newS[j];
            }
        }
        hashLong[hl[0]] = newL[0]; hashSmall[hs[0]] = newS[0];
        uint32_t idxl[NP + 1], idxs[NP], pl[NP + 1]; bool psv[NP];
#pragma unroll
        for (int k = 0; k <= NP; k++) { idxl[k] = ZE_CELL_IDX(cL[k]); pl[k] = (idxl[k] >= 2 && (cL[k] >> PB) == (ZE_TL(pL[k]) >> PB)) ? 1u : 0u; }
#pragma unroll
        for (int k = 0; k < NP; k++) { idxs[k] = ZE_CELL_IDX(cS[k]); psv[k] = idxs[k] >= 2 && (cS[k] >> PB) == (ZE_TS(w[k]) >> PB); }
        if (!fresh) pl[0] = pl0;
        // round 2: the bytes of the plausible candidates (the others read one address the whole wave shares)
        uint64_t xl[NP + 1]; uint32_t cs[NP];
#pragma unroll
        for (int k = 0; k <= NP; k++) xl[k] = zh_ld64((pl[k] && (k > 0 || fresh)) ? src + (idxl[k] - 2) : idle);
#pragma unroll
        for (int k = 0; k < NP; k++) cs[k] = zh_ld32(psv[k] ? src + (idxs[k] - 2) : idle);
#pragma unroll
        for (int k = 0; k <= NP; k++) xl[k] = zh_opaque64(xl[k]);                 // no load sinks into a branch
#pragma unroll
        for (int k = 0; k < NP; k++) cs[k] = zh_opaque(cs[k]);
        if (!fresh) xl[0] = cl0;
        int fnd[NP];
#pragma unroll
        for (int k = 0; k < NP; k++)
            fnd[k] = (off1 > 0 && rp[k] == (uint32_t)(w[k] >> 8)) ? 1 : (pl[k] && xl[k] == w[k]) ? 2 : (psv[k] && cs[k] == (uint32_t)w[k]) ? 3 : 0;
        uint32_t P = NP; int found = 0;                                            // the first real probe that matched
#pragma unroll
        for (int k = NP - 1; k >= 0; k--) if ((uint32_t)k < R && fnd[k]) { P = (uint32_t)k; found = fnd[k]; }
        // the table writes of the probes that really happened: 1 .. P (a match at P) or 1 .. R - 1 (none), in the reference's order
#pragma unroll
        for (int k = 1; k < NP; k++) if ((uint32_t)k <= P && (uint32_t)k < R) { hashLong[hl[k]] = newL[k]; hashSmall[hs[k]] = newS[k]; }
        if (found) {
            ZE_STAT(11);
            uint32_t ipP = pos[0], ipP1 = pos[1], stepP = st[0], idxlP = idxl[0], idxsP = idxs[0], idxl1 = idxl[1], pl1 = pl[1], hl1 = hl[1], newL1 = newL[1];
            uint64_t w1 = w[1], cl1 = xl[1];
#pragma unroll
            for (int k = 1; k < NP; k++) if (P == (uint32_t)k) { ipP = pos[k]; ipP1 = pos[k + 1]; stepP = st[k]; idxlP = idxl[k]; idxsP = idxs[k]; idxl1 = idxl[k + 1]; pl1 = pl[k + 1]; hl1 = hl[k + 1]; newL1 = newL[k + 1]; w1 = w[k + 1]; cl1 = xl[k + 1]; }
            uint32_t ipm = ipP, mpos = 0, ca, cb, add;
            if (found == 1) { ipm = ipP + 1; ca = ipP + 5; cb = ipP + 5 - off1; add = 4; }
            else if (found == 2) { mpos = idxlP - 2; ca = ipP + 8; cb = mpos + 8; add = 8; }
            else { mpos = idxsP - 2; ca = ipP + 4; cb = mpos + 4; add = 4; }
            uint32_t mLength = ze_count_fwd(src, ca, cb, srcSize) + add;
            if (found == 3 && pl1 && idxl1 > 2 && cl1 == w1) {          // a long match one step ahead beats a shorter short match (zstd.c:31192-31201)
                const uint32_t m1 = idxl1 - 2;
                const uint32_t l1 = ze_count_fwd(src, ipP1 + 8, m1 + 8, srcSize) + 8;
                if (l1 > mLength) { ipm = ipP1; mLength = l1; mpos = m1; }
            }
            uint32_t offBase = 1;
            if (found >= 2) {
                const uint32_t offset = ipm - mpos;
                while (ipm > anchor && mpos > 0) {                      // catch up (zstd.c:31182, :31204), 8 bytes a round
                    ZE_STAT(13);
                    const uint32_t room = ipm - anchor < mpos ? ipm - anchor : mpos;
                    if (mpos >= 8) {
                        const uint64_t d = zh_ld64(src + ipm - 8) ^ zh_ld64(src + mpos - 8);
                        uint32_t k = d ? (uint32_t)(zh_clz64(d) >> 3) : 8u;
                        if (k > room) k = room;
                        ipm -= k; mpos -= k; mLength += k;
                        if (k < 8) break;
                    } else {
                        if (src[ipm - 1] != src[mpos - 1]) break;
                        ipm--; mpos--; mLength++;
                    }
                }
                off2 = off1; off1 = offset;
                if (stepP < 4) hashLong[hl1] = newL1;
                offBase = offset + 3;
            }
            seqs[nseq++] = ZE_SEQ_PACK(offBase, ipm - anchor, mLength);
            const uint32_t pI = ipP + 2;
            ip = ipm + mLength; anchor = ip;
            if (ip <= ilimit) {
                const uint64_t wI = zh_ld64(src + pI), wE2 = zh_ld64(src + ip - 2), wE1 = zh_ld64(src + ip - 1);
                uint64_t wr = zh_ld64(src + ip); uint32_t r2 = zh_ld32(src + ip - off2);
                const uint32_t qI = ZE_PL(wI), qE = ZE_PL(wE2);
                hashLong[qI >> shL] = (pI + 2) | ZE_TL(qI);
                hashLong[qE >> shL] = ip | ZE_TL(qE);
                hashSmall[ZE_PS(wI) >> shS] = (pI + 2) | ZE_TS(wI);
                hashSmall[ZE_PS(wE1) >> shS] = (ip + 1) | ZE_TS(wE1);
                while (off2 > 0 && (uint32_t)wr == r2) {                // immediate repeat-offset matches (zstd.c:31236-31250)
                    ZE_STAT(14);
                    const uint32_t r = ze_count_fwd(src, ip + 4, ip + 4 - off2, srcSize) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    const uint32_t qr = ZE_PL(wr);
                    hashSmall[ZE_PS(wr) >> shS] = (ip + 2) | ZE_TS(wr);
                    hashLong[qr >> shL] = (ip + 2) | ZE_TL(qr);
                    seqs[nseq++] = ZE_SEQ_PACK(1, 0, r);
                    ip += r; anchor = ip;
                    if (ip > ilimit) break;
                    wr = zh_ld64(src + ip); r2 = zh_ld32(src + ip - off2);
                }
            }
            fresh = true;
        } else {                                                        // every real probe failed: go on from pos[R] with its long cell and candidate in hand
            ip = pos[1]; step = st[1]; nextStep = nx[1]; cellL0 = cL[1]; pl0 = pl[1]; cl0 = xl[1];
#pragma unroll
            for (int k = 2; k <= NP; k++) if (R == (uint32_t)k) { ip = pos[k]; step = st[k]; nextStep = nx[k]; cellL0 = cL[k]; pl0 = pl[k]; cl0 = xl[k]; }
            fresh = false;
        }
    }
#undef ZE_PL
#undef ZE_PS
#undef ZE_TL
#undef ZE_TS
#undef ZE_CELL_IDX
#define ZE_CELL_IDX(c) ((c) & 0x3FFFFu)
    if (BLK) {                                                        // zstd.c:31252-31258
        saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
        rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
    }
    return nseq;
}
template <int PB, bool BLK>
ZH_DEV uint32_t ze_dfast_flat_t(uint64_t* seqs, const uint8_t* src, uint32_t bs, uint32_t be, int hlog, int clog, int mml, uint32_t* hashLong, uint32_t* hashSmall, uint32_t* rep, const uint8_t* idle = nullptr)
{
    return ze_dfast_flat_np<PB, BLK, 2>(seqs, src, bs, be, hlog, clog, mml, hashLong, hashSmall, rep, idle ? idle : src);
}
template <int NP>
ZH_DEV uint32_t ze_dfast_flatn(uint64_t* seqs, const uint8_t* src, uint32_t srcSize, int hlog, int clog, int mml, uint32_t* hashLong, uint32_t* hashSmall, const uint8_t* idle = nullptr, uint32_t epoch = 0)
{
    return ze_dfast_flat_np<18, false, NP>(seqs, src, 0, srcSize, hlog, clog, mml, hashLong, hashSmall, nullptr, idle ? idle : src, epoch);
}
ZH_DEV uint32_t ze_dfast_flat4(uint64_t* seqs, const uint8_t* src, uint32_t srcSize, int hlog, int clog, int mml, uint32_t* hashLong, uint32_t* hashSmall, const uint8_t* idle = nullptr, uint32_t epoch = 0)
{
    return ze_dfast_flatn<4>(seqs, src, srcSize, hlog, clog, mml, hashLong, hashSmall, idle, epoch);
}
ZH_DEV uint32_t ze_dfast_flat(uint64_t* seqs, const uint8_t* src, uint32_t srcSize, int hlog, int clog, int mml, uint32_t* hashLong, uint32_t* hashSmall, const uint8_t* idle = nullptr, uint32_t epoch = 0)
{
    return ze_dfast_flat_np<18, false, 2>(seqs, src, 0, srcSize, hlog, clog, mml, hashLong, hashSmall, nullptr, idle ? idle : src, epoch);
}


// ------------------------------------------------------------------------------------------ fast strategy (levels 1-2, negative levels)
// ZSTD_compressBlock_fast_noDict_generic (zstd.c:31906) for a block that is the whole frame: one hash table of hashLog bits over
// minMatch bytes, cells hold position + 2. Positions are examined in pairs `step` apart (step grows by one per 128 bytes without a
// match), with a repcode test two positions ahead of a pair's first; after a hit at a pair's second position the pending table
// write is kept only while step <= 4. One lane.
ZH_DEVFN uint32_t ze_fast_g(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* frame, const uint8_t* src, uint32_t srcSize, const ZePar& cp,
                            uint32_t* table, uint32_t* rep)
{
    const int hlog = cp.hlog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t stepSize = (uint32_t)cp.tlen + (cp.tlen == 0) + 1;
    const uint8_t* const base = frame - 2;
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    const uint32_t maxDist = 1u << cp.wlog;
    const uint32_t endIndex = (uint32_t)(iend - base);
    const uint32_t LOW = endIndex - 2 > maxDist ? endIndex - maxDist : 2;
    const uint8_t* anchor = src;
    const uint8_t* ip0 = src + ((uint32_t)(src - base) == LOW ? 1 : 0);      // empty prefix (zstd.c:31958 `ip0 += (ip0 == prefixStart)`), see ze_dfast_g
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    {   const uint32_t c0 = (uint32_t)(ip0 - base); const uint32_t maxRep = c0 - 2 > maxDist ? maxDist : c0 - 2;
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; } }
    uint32_t nseq = 0; uint8_t* lp = lits;
    // (lits == nullptr: sequences only -- the entropy kernel gathers the literals from the source, wave-parallel, as for the flat search; this lane's byte-by-byte copy of ~45 KB per
    // 128 KiB source was a good part of the lane-serial kernel's time on fast-strategy batches)
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); if (lits) { for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = anchor[i_]; lp += ll_; } \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
#define ZE_IDX(p) ((uint32_t)((p) - base))
    if (srcSize >= 8) for (;;) {
        uint32_t step = stepSize;
        const uint8_t* nextStep = ip0 + 128;
        const uint8_t* ip1 = ip0 + 1; const uint8_t* ip2 = ip0 + step; const uint8_t* ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        uint32_t hash0 = ze_hash(ip0, hlog, mls), hash1 = ze_hash(ip1, hlog, mls);
        uint32_t matchIdx = table[hash0];
        uint32_t current0 = 0, offBase = 0, mLength = 0; const uint8_t* match0 = ip0;
        int found = 0;
        do {
            const uint32_t rval = zh_ld32(ip2 - rep1);
            current0 = ZE_IDX(ip0); table[hash0] = current0;
            if (zh_ld32(ip2) == rval && rep1 > 0) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = ip0[-1] == match0[-1] ? 1u : 0u;
                ip0 -= mLength; match0 -= mLength;
                offBase = 1; mLength += 4;
                table[hash1] = ZE_IDX(ip1);
                found = 1; break;
            }
            if (matchIdx >= LOW && zh_ld32(base + matchIdx) == zh_ld32(ip0)) { table[hash1] = ZE_IDX(ip1); found = 2; break; }
            matchIdx = table[hash1];
            hash0 = hash1; hash1 = ze_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = ZE_IDX(ip0); table[hash0] = current0;
            if (matchIdx >= LOW && zh_ld32(base + matchIdx) == zh_ld32(ip0)) { if (step <= 4) table[hash1] = ZE_IDX(ip1); found = 2; break; }
            matchIdx = table[hash1];
            hash0 = hash1; hash1 = ze_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;
        if (found == 2) {
            match0 = base + matchIdx;
            rep2 = rep1; rep1 = (uint32_t)(ip0 - match0);
            offBase = rep1 + 3; mLength = 4;
            while (ip0 > anchor && match0 > base + LOW && ip0[-1] == match0[-1]) { ip0--; match0--; mLength++; }
        }
        mLength += ze_common_len(ip0 + mLength, match0 + mLength, iend);
        ZE_STORE(ip0 - anchor, offBase, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {
            table[ze_hash(base + current0 + 2, hlog, mls)] = current0 + 2;
            table[ze_hash(ip0 - 2, hlog, mls)] = ZE_IDX(ip0 - 2);
            if (rep2 > 0) {
                while (ip0 <= ilimit && zh_ld32(ip0) == zh_ld32(ip0 - rep2)) {
                    const uint32_t rLength = ze_common_len(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    const uint32_t t = rep2; rep2 = rep1; rep1 = t;
                    table[ze_hash(ip0, hlog, mls)] = ZE_IDX(ip0);
                    ZE_STORE(0, 1, rLength);
                    ip0 += rLength; anchor = ip0;
                }
            }
        }
    }
#undef ZE_IDX
#undef ZE_STORE
    if (lits) { const uint32_t lastLL = (uint32_t)(iend - anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = anchor[i]; lp += lastLL; }
    *pLit = lits ? (uint32_t)(lp - lits) : 0u;
    if (saved1 != 0 && rep1 != 0) saved2 = saved1;
    rep[0] = rep1 ? rep1 : saved1; rep[1] = rep2 ? rep2 : saved2;
    return nseq;
}
ZH_DEV uint32_t ze_fast(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* src, uint32_t srcSize, const ZePar& cp, uint32_t* table)
{
    uint32_t rep[2] = {1, 4};
    return ze_fast_g(seqs, lits, pLit, src, src, srcSize, cp, table, rep);
}

// ------------------------------------------------------------------------------------------ double-fast search against an attached dictionary
// ZSTD_compressBlock_doubleFast_dictMatchState_generic (zstd.c:31262) for a frame of one block. One index space: dictionary
// content byte k is index 2 + k, the source follows at CE = 2 + contentSize, so offsets are index differences. The frame's own
// tables hold plain indices; the dictionary's hold (index << 8 | tag) (ZSTD_writeTaggedIndex, zstd.c:20636). One lane.
struct ZeSpace { const uint8_t* content; const uint8_t* src; uint32_t CE, end; };
ZH_DEV uint32_t ze_sp_byte(const ZeSpace& sp, uint32_t i) { return i < sp.CE ? sp.content[i - 2] : sp.src[i - sp.CE]; }
ZH_DEV uint32_t ze_sp_rd32(const ZeSpace& sp, uint32_t i)
{
    if (i >= sp.CE) return zh_ld32(sp.src + (i - sp.CE));
    if (i + 4 <= sp.CE) return zh_ld32(sp.content + (i - 2));
    uint32_t v = 0; for (uint32_t k = 0; k < 4; k++) v |= ze_sp_byte(sp, i + k) << (8 * k);
    return v;
}
ZH_DEV uint64_t ze_sp_rd64(const ZeSpace& sp, uint32_t i)
{
    if (i >= sp.CE) return zh_ld64(sp.src + (i - sp.CE));
    if (i + 8 <= sp.CE) return zh_ld64(sp.content + (i - 2));
    uint64_t v = 0; for (uint32_t k = 0; k < 8; k++) v |= (uint64_t)ze_sp_byte(sp, i + k) << (8 * k);
    return v;
}
// ZSTD_count_2segments (zstd.c:20034): the match may run off the dictionary's end into the start of the source
ZH_DEV uint32_t ze_sp_count(const ZeSpace& sp, uint32_t ip, uint32_t m)
{
    const uint8_t* const a = sp.src + (ip - sp.CE);
    const uint8_t* const iend = sp.src + (sp.end - sp.CE);
    if (m >= sp.CE) return ze_common_len(a, sp.src + (m - sp.CE), iend);
    const uint32_t inDict = sp.CE - m, room = sp.end - ip;
    const uint8_t* const vend = a + (inDict < room ? inDict : room);
    const uint32_t n = ze_common_len(a, sp.content + (m - 2), vend);
    if (n != inDict) return n;
    return n + ze_common_len(a + n, sp.src, iend);
}

ZH_DEVFN uint32_t ze_dfast_dict(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* src, uint32_t srcSize, const ZePar& cp,
                                const ZeCDict& cd, const uint8_t* content, const uint32_t* dHashLong, const uint32_t* dHashSmall,
                                uint32_t* hashLong, uint32_t* hashSmall)
{
    const int hl = cp.hlog, hs = cp.clog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t CE = 2 + cd.contentSize, DS = 2;
    ZeSpace sp; sp.content = content; sp.src = src; sp.CE = CE; sp.end = CE + srcSize;
    const int dhl = cd.hlog + 8, dhs = cd.clog + 8;
    const uint32_t iend = CE + srcSize;
    uint32_t ip = CE, anchor = CE;
    uint32_t off1 = cd.rep[0], off2 = cd.rep[1];
    uint32_t nseq = 0; uint8_t* lp = lits;
#define ZE_SRC(i) (src + ((i) - CE))
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = a_[i_]; lp += ll_; \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
#define ZE_BACK(LOW) while (ip > anchor && m > (LOW) && ze_sp_byte(sp, ip - 1) == ze_sp_byte(sp, m - 1)) { ip--; m--; mLength++; }
    if (srcSize >= 8) {
        const uint32_t ilimit = iend - 8;
        while (ip < ilimit) {
            uint32_t mLength = 0, offset = 0;
            const uint32_t h2 = ze_hash(ZE_SRC(ip), hl, 8), h = ze_hash(ZE_SRC(ip), hs, mls);
            const uint32_t dTagL = ze_hash(ZE_SRC(ip), dhl, 8), dTagS = ze_hash(ZE_SRC(ip), dhs, mls);
            const uint32_t dEntL = dHashLong[dTagL >> 8], dEntS = dHashSmall[dTagS >> 8];
            const bool tagL = (dEntL & 255) == (dTagL & 255), tagS = (dEntS & 255) == (dTagS & 255);
            const uint32_t curr = ip;
            const uint32_t mIdxL = hashLong[h2]; uint32_t mIdxS = hashSmall[h];
            const uint32_t repIndex = curr + 1 - off1;
            hashLong[h2] = curr; hashSmall[h] = curr;
            int found = 0;
            if ((uint32_t)((CE - 1) - repIndex) >= 3 && ze_sp_rd32(sp, repIndex) == zh_ld32(ZE_SRC(ip + 1))) {
                mLength = ze_sp_count(sp, ip + 1 + 4, repIndex + 4) + 4;
                ip++;
                ZE_STORE(ip - anchor, 1, mLength);
                found = 1;
            } else {
                if (mIdxL >= CE && zh_ld64(ZE_SRC(mIdxL)) == zh_ld64(ZE_SRC(ip))) {
                    uint32_t m = mIdxL;
                    mLength = ze_sp_count(sp, ip + 8, m + 8) + 8;
                    offset = ip - m;
                    ZE_BACK(CE)
                    found = 2;
                } else if (tagL) {
                    uint32_t m = dEntL >> 8;
                    if (m > DS && ze_sp_rd64(sp, m) == zh_ld64(ZE_SRC(ip))) {
                        mLength = ze_sp_count(sp, ip + 8, m + 8) + 8;
                        offset = curr - m;
                        ZE_BACK(DS)
                        found = 2;
                    }
                }
                if (!found) {
                    bool shortCand = false; uint32_t match = 0;
                    if (mIdxS > CE) { if (zh_ld32(ZE_SRC(mIdxS)) == zh_ld32(ZE_SRC(ip))) { shortCand = true; match = mIdxS; } }
                    else if (tagS) {
                        match = dEntS >> 8; mIdxS = match;
                        if (match > DS && ze_sp_rd32(sp, match) == zh_ld32(ZE_SRC(ip))) shortCand = true;
                    }
                    if (!shortCand) { ip += ((ip - anchor) >> 8) + 1; continue; }
                    {   // a short match exists: a long match one position later wins
                        const uint32_t hl3 = ze_hash(ZE_SRC(ip + 1), hl, 8), dTagL3 = ze_hash(ZE_SRC(ip + 1), dhl, 8);
                        const uint32_t mIdxL3 = hashLong[hl3], dEntL3 = dHashLong[dTagL3 >> 8];
                        const bool tagL3 = (dEntL3 & 255) == (dTagL3 & 255);
                        hashLong[hl3] = curr + 1;
                        if (mIdxL3 >= CE && zh_ld64(ZE_SRC(mIdxL3)) == zh_ld64(ZE_SRC(ip + 1))) {
                            uint32_t m = mIdxL3;
                            mLength = ze_sp_count(sp, ip + 9, m + 8) + 8;
                            ip++;
                            offset = ip - m;
                            ZE_BACK(CE)
                            found = 2;
                        } else if (tagL3) {
                            uint32_t m = dEntL3 >> 8;
                            if (m > DS && ze_sp_rd64(sp, m) == zh_ld64(ZE_SRC(ip + 1))) {
                                mLength = ze_sp_count(sp, ip + 1 + 8, m + 8) + 8;
                                ip++;
                                offset = curr + 1 - m;
                                ZE_BACK(DS)
                                found = 2;
                            }
                        }
                    }
                    if (!found) {
                        uint32_t m = match;
                        mLength = ze_sp_count(sp, ip + 4, m + 4) + 4;
                        offset = curr - mIdxS;
                        if (mIdxS < CE) { ZE_BACK(DS) } else { ZE_BACK(CE) }
                        found = 2;
                    }
                }
                off2 = off1; off1 = offset;
                ZE_STORE(ip - anchor, offset + 3, mLength);
            }
            ip += mLength; anchor = ip;
            if (ip <= ilimit) {
                const uint32_t ins = curr + 2;
                hashLong[ze_hash(ZE_SRC(ins), hl, 8)] = ins;
                hashLong[ze_hash(ZE_SRC(ip - 2), hl, 8)] = ip - 2;
                hashSmall[ze_hash(ZE_SRC(ins), hs, mls)] = ins;
                hashSmall[ze_hash(ZE_SRC(ip - 1), hs, mls)] = ip - 1;
                while (ip <= ilimit) {
                    const uint32_t rep2 = ip - off2;
                    if (!((uint32_t)((CE - 1) - rep2) >= 3 && ze_sp_rd32(sp, rep2) == zh_ld32(ZE_SRC(ip)))) break;
                    const uint32_t r = ze_sp_count(sp, ip + 4, rep2 + 4) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    ZE_STORE(0, 1, r);
                    hashSmall[ze_hash(ZE_SRC(ip), hs, mls)] = ip;
                    hashLong[ze_hash(ZE_SRC(ip), hl, 8)] = ip;
                    ip += r; anchor = ip;
                }
            }
        }
    }
#undef ZE_BACK
#undef ZE_STORE
    {   const uint32_t lastLL = iend - anchor; const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = a_[i]; lp += lastLL; }
#undef ZE_SRC
    *pLit = (uint32_t)(lp - lits);
    return nseq;
}

// ------------------------------------------------------------------------------------------ double-fast + attached dictionary, FLAT
// The search of ze_dfast_dict above for the batch shape dictionaries exist for (BASELINE configs[3]: hundreds of thousands of small
// documents), the way ze_dfast_flat re-expressed the dictionary-less one: one LANE per document, 64 documents per wave, every document of
// the chunk in flight, ONE loop whose trip examines one probe position in three memory rounds whatever the lanes decide --
//   round 0: the probe's own bytes (position ip and ip + 1) and the repeat-offset candidate's;
//   round 1: every table cell the reference could consult for this position: its own long / short cell, the dictionary's tagged long /
//            short cell, and -- speculatively, they are only reads -- the long cells of position ip + 1 (own and dictionary), which the
//            reference consults after a short candidate was found (zstd.c:31385-31420);
//   round 2: the bytes of every candidate that is PLAUSIBLE (own cell in use; dictionary cell carrying the probe's 8-bit tag);
// then the reference's decision order over what was fetched (repeat offset, own long, dictionary long, short, long at ip + 1). All loads of
// a round are unconditional with always-valid addresses, so a trip is three round trips instead of the up to six dependent ones of the
// nested-loop form, and lanes wait for one another only in the match epilogue (count / catch-up / insertions), never to FIND a match.
// The table writes are the reference's, in its order: own cells of ip always (zstd.c:31308), the long cell of ip + 1 only on the path
// that consults it (:31391). Sequences only -- the entropy stage gathers the literals from the source (ze_gather_literals).
// Index space as in ze_dfast_dict: dictionary content byte k is index 2 + k, the source starts at CE = 2 + contentSize.
ZH_DEV uint32_t ze_dfast_dict_flat(uint64_t* seqs, const uint8_t* src, uint32_t srcSize, const ZePar& cp, const ZeCDict& cd, const uint8_t* content,
                                   const uint32_t* dHashLong, const uint32_t* dHashSmall, uint32_t* hashLong, uint32_t* hashSmall, uint32_t epoch = 0, uint32_t epochShift = 31)
{
    // own cells: index | launch number << ES (ZhipEncodeArgs.tabEpoch); a cell of another launch -- or, with epoch 0, a zeroed one -- reads as 0 = empty (own candidates are >= CE)
    const uint32_t ES = epoch ? epochShift : 31u, EW = epoch << ES, IM = (1u << ES) - 1u;
#define ZE_OWN_RD(x) ((((x) >> ES) == epoch) ? ((x) & IM) : 0u)
#define ZE_OWN_WR(i) ((i) | EW)
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t shL = 64u - (uint32_t)cp.hlog, shDL = 64u - (uint32_t)(cd.hlog + 8);
    const uint32_t shlS = mls == 4 ? 32u : (uint32_t)(64 - 8 * mls);
    const uint64_t primeS = mls == 4 ? 2654435761ull : mls == 5 ? 889523592379ull : mls == 6 ? 227718039650203ull : 58295818150454627ull;
    // (for mls == 4 the reference hashes 32 bits: (u32 * prime) >> (32 - bits) == ((u << 32) * prime) >> (64 - bits))
    const uint32_t shS = 64u - (uint32_t)cp.clog, shDS = 64u - (uint32_t)(cd.clog + 8);
    const uint32_t CE = 2 + cd.contentSize, DS = 2;
    ZeSpace sp; sp.content = content; sp.src = src; sp.CE = CE; sp.end = CE + srcSize;
    const uint32_t iend = CE + srcSize;
    uint32_t ip = CE, anchor = CE, off1 = cd.rep[0], off2 = cd.rep[1], nseq = 0;
    if (srcSize < 8) return 0;
    const uint32_t ilimit = iend - 8;
#define ZE_SRC(i) (src + ((i) - CE))
    // 8 readable bytes at index i of the index space, or -- when they straddle the end of the dictionary's content -- the probe's own (the
    // caller re-reads such a candidate byte by byte: ze_sp_rd64)
#define ZE_SPP(i, fallback) ((i) >= CE ? ZE_SRC(i) : (i) + 8 <= CE ? content + ((i) - 2) : (fallback))
#define ZE_BACK(LOW) while (ip > anchor && m > (LOW) && ze_sp_byte(sp, ip - 1) == ze_sp_byte(sp, m - 1)) { ip--; m--; mLength++; }
    while (ip < ilimit) {
        const uint32_t curr = ip;
        const uint8_t* const own = ZE_SRC(ip);
        // ---- round 0
        const uint64_t w = zh_ld64(own), w1 = zh_ld64(own + 1);
        const uint32_t repIndex = curr + 1 - off1;
        const bool repOK = (uint32_t)((CE - 1) - repIndex) >= 3;                // not across the dictionary / source boundary (zstd.c:31312)
        const uint32_t rp = zh_ld32(repOK ? (repIndex >= CE ? ZE_SRC(repIndex) : content + (repIndex - 2)) : own);
        const uint64_t pL = w * 0xCF1BBCDCB7A56463ull, pL1 = w1 * 0xCF1BBCDCB7A56463ull, pS = (w << shlS) * primeS;
        const uint32_t h2 = (uint32_t)(pL >> shL), h = (uint32_t)(pS >> shS), hl3 = (uint32_t)(pL1 >> shL);
        const uint32_t dTagL = (uint32_t)(pL >> shDL), dTagS = (uint32_t)(pS >> shDS), dTagL3 = (uint32_t)(pL1 >> shDL);
        // ---- round 1
        const uint32_t rawL = hashLong[h2], rawS = hashSmall[h], dEntL = dHashLong[dTagL >> 8], dEntS = dHashSmall[dTagS >> 8];
        const uint32_t rawL3 = hashLong[hl3]; const uint32_t dEntL3 = dHashLong[dTagL3 >> 8];
        const uint32_t mIdxL = ZE_OWN_RD(rawL), mIdxS0 = ZE_OWN_RD(rawS);
        uint32_t mIdxL3 = ZE_OWN_RD(rawL3);
        if (hl3 == h2) mIdxL3 = curr;                                           // what the reference's later read sees after its write below
        hashLong[h2] = ZE_OWN_WR(curr); hashSmall[h] = ZE_OWN_WR(curr);
        const bool tagL = (dEntL & 255) == (dTagL & 255), tagS = (dEntS & 255) == (dTagS & 255), tagL3 = (dEntL3 & 255) == (dTagL3 & 255);
        const uint32_t dL = dEntL >> 8, dS = dEntS >> 8, dL3 = dEntL3 >> 8;
        const bool ownL = mIdxL >= CE, ownS = mIdxS0 > CE, ownL3 = mIdxL3 >= CE;
        const bool dctL = !ownL && tagL && dL > DS, dctS = !ownS && tagS && dS > DS, dctL3 = tagL3 && dL3 > DS;     // (dctL: consulted only when the own cell fails -- see below)
        // ---- round 2: candidates' bytes
        uint64_t cL = zh_ld64(ownL ? ZE_SRC(mIdxL) : own);
        uint64_t cDL = zh_ld64((tagL && dL > DS) ? ZE_SPP(dL, own) : own);
        uint32_t cS = zh_ld32(ownS ? ZE_SRC(mIdxS0) : dctS ? ZE_SPP(dS, own) : own);
        uint64_t cL3 = zh_ld64(ownL3 ? ZE_SRC(mIdxL3) : own + 1);
        uint64_t cDL3 = zh_ld64(dctL3 ? ZE_SPP(dL3, own + 1) : own + 1);
        cL = zh_opaque64(cL); cDL = zh_opaque64(cDL); cS = zh_opaque(cS); cL3 = zh_opaque64(cL3); cDL3 = zh_opaque64(cDL3);     // no load sinks into a branch
        (void)dctL;
        // candidates across the end of the dictionary's content (a few per batch): byte by byte
        if (tagL && dL > DS && dL < CE && dL + 8 > CE) cDL = ze_sp_rd64(sp, dL);
        if (dctS && dS < CE && dS + 8 > CE) cS = ze_sp_rd32(sp, dS);
        if (dctL3 && dL3 < CE && dL3 + 8 > CE) cDL3 = ze_sp_rd64(sp, dL3);
        // ---- the reference's decision order (zstd.c:31312-31440)
        uint32_t mLength = 0, offset = 0;
        int found = 0;
        if (repOK && rp == (uint32_t)(w >> 8)) {
            mLength = ze_sp_count(sp, ip + 1 + 4, repIndex + 4) + 4;
            ip++;
            seqs[nseq++] = ZE_SEQ_PACK(1, ip - anchor, mLength);
            found = 1;
        } else {
            if (ownL && cL == w) {
                uint32_t m = mIdxL;
                mLength = ze_sp_count(sp, ip + 8, m + 8) + 8;
                offset = ip - m;
                ZE_BACK(CE)
                found = 2;
            } else if (tagL && dL > DS && cDL == w) {
                uint32_t m = dL;
                mLength = ze_sp_count(sp, ip + 8, m + 8) + 8;
                offset = curr - m;
                ZE_BACK(DS)
                found = 2;
            }
            if (!found) {
                const bool shortCand = (ownS || dctS) && cS == (uint32_t)w;
                if (!shortCand) { ip += ((ip - anchor) >> 8) + 1; continue; }
                const uint32_t mIdxS = ownS ? mIdxS0 : dS;
                hashLong[hl3] = ZE_OWN_WR(curr + 1);                            // the long table is consulted -- and written -- one position ahead
                if (ownL3 && cL3 == w1) {
                    uint32_t m = mIdxL3;
                    mLength = ze_sp_count(sp, ip + 9, m + 8) + 8;
                    ip++;
                    offset = ip - m;
                    ZE_BACK(CE)
                } else if (dctL3 && cDL3 == w1) {
                    uint32_t m = dL3;
                    mLength = ze_sp_count(sp, ip + 1 + 8, m + 8) + 8;
                    ip++;
                    offset = curr + 1 - m;
                    ZE_BACK(DS)
                } else {
                    uint32_t m = mIdxS;
                    mLength = ze_sp_count(sp, ip + 4, m + 4) + 4;
                    offset = curr - mIdxS;
                    if (mIdxS < CE) { ZE_BACK(DS) } else { ZE_BACK(CE) }
                }
                found = 2;
            }
            off2 = off1; off1 = offset;
            seqs[nseq++] = ZE_SEQ_PACK(offset + 3, ip - anchor, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = curr + 2;
            const uint64_t wI = zh_ld64(ZE_SRC(ins)), wE2 = zh_ld64(ZE_SRC(ip - 2)), wE1 = zh_ld64(ZE_SRC(ip - 1));
            hashLong[(uint32_t)((wI * 0xCF1BBCDCB7A56463ull) >> shL)] = ZE_OWN_WR(ins);
            hashLong[(uint32_t)((wE2 * 0xCF1BBCDCB7A56463ull) >> shL)] = ZE_OWN_WR(ip - 2);
            hashSmall[(uint32_t)(((wI << shlS) * primeS) >> shS)] = ZE_OWN_WR(ins);
            hashSmall[(uint32_t)(((wE1 << shlS) * primeS) >> shS)] = ZE_OWN_WR(ip - 1);
            while (ip <= ilimit) {
                const uint32_t rep2 = ip - off2;
                if (!((uint32_t)((CE - 1) - rep2) >= 3 && ze_sp_rd32(sp, rep2) == zh_ld32(ZE_SRC(ip)))) break;
                const uint32_t r = ze_sp_count(sp, ip + 4, rep2 + 4) + 4;
                const uint32_t t = off2; off2 = off1; off1 = t;
                seqs[nseq++] = ZE_SEQ_PACK(1, 0, r);
                const uint64_t wr = zh_ld64(ZE_SRC(ip));
                hashSmall[(uint32_t)(((wr << shlS) * primeS) >> shS)] = ZE_OWN_WR(ip);
                hashLong[(uint32_t)((wr * 0xCF1BBCDCB7A56463ull) >> shL)] = ZE_OWN_WR(ip);
                ip += r; anchor = ip;
            }
        }
    }
#undef ZE_OWN_RD
#undef ZE_OWN_WR
#undef ZE_BACK
#undef ZE_SPP
#undef ZE_SRC
    return nseq;
}

// ------------------------------------------------------------------------------------------ fast search against an attached dictionary
// ZSTD_compressBlock_fast_dictMatchState_generic (zstd.c:32197) for a frame of one block, in the index space of ze_dfast_dict (dictionary
// content byte k is index 2 + k, the source starts at CE). The dictionary has ONE tagged table here (ZSTD_fillHashTableForCDict). One lane.
ZH_DEVFN uint32_t ze_fast_dict(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* src, uint32_t srcSize, const ZePar& cp,
                               const ZeCDict& cd, const uint8_t* content, const uint32_t* dHash, uint32_t* hashTable)
{
    const int hlog = cp.hlog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t stepSize = (uint32_t)cp.tlen + (cp.tlen == 0);
    const uint32_t CE = 2 + cd.contentSize, DS = 2;
    ZeSpace sp; sp.content = content; sp.src = src; sp.CE = CE; sp.end = CE + srcSize;
    const int dhb = cd.hlog + 8;
    const uint32_t iend = CE + srcSize;
    uint32_t ip0 = CE, ip1 = CE + stepSize, anchor = CE;
    uint32_t off1 = cd.rep[0], off2 = cd.rep[1];
    uint32_t nseq = 0; uint8_t* lp = lits;
#define ZE_SRC(i) (src + ((i) - CE))
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = a_[i_]; lp += ll_; \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
    if (srcSize >= 8) {
        const uint32_t ilimit = iend - 8;
        while (ip1 <= ilimit) {
            uint32_t mLength = 0;
            uint32_t hash0 = ze_hash(ZE_SRC(ip0), hlog, mls);
            const uint32_t dTag0 = ze_hash(ZE_SRC(ip0), dhb, mls);
            uint32_t dEnt = dHash[dTag0 >> 8];
            bool dictTagsMatch = (dEnt & 255) == (dTag0 & 255);
            uint32_t matchIndex = hashTable[hash0];
            uint32_t curr = ip0;
            uint32_t step = stepSize;
            uint32_t nextStep = ip0 + 256;                                 // kStepIncr = 1 << kSearchStrength
            bool done = false;
            for (;;) {
                const uint32_t repIndex = curr + 1 - off1;
                const uint32_t hash1 = ze_hash(ZE_SRC(ip1), hlog, mls);
                const uint32_t dTag1 = ze_hash(ZE_SRC(ip1), dhb, mls);
                hashTable[hash0] = curr;
                if ((uint32_t)((CE - 1) - repIndex) >= 3 && ze_sp_rd32(sp, repIndex) == zh_ld32(ZE_SRC(ip0 + 1))) {
                    mLength = ze_sp_count(sp, ip0 + 1 + 4, repIndex + 4) + 4;
                    ip0++;
                    ZE_STORE(ip0 - anchor, 1, mLength);
                    break;
                }
                if (dictTagsMatch) {
                    uint32_t m = dEnt >> 8;
                    // "to replicate extDict parse behavior, we only use dict matches when the normal matchIndex is invalid"
                    if (m > DS && ze_sp_rd32(sp, m) == zh_ld32(ZE_SRC(ip0)) && matchIndex <= CE) {
                        const uint32_t offset = curr - m;
                        mLength = ze_sp_count(sp, ip0 + 4, m + 4) + 4;
                        while (ip0 > anchor && m > DS && ze_sp_byte(sp, ip0 - 1) == ze_sp_byte(sp, m - 1)) { ip0--; m--; mLength++; }
                        off2 = off1; off1 = offset;
                        ZE_STORE(ip0 - anchor, offset + 3, mLength);
                        break;
                    }
                }
                if (matchIndex >= CE && zh_ld32(ZE_SRC(matchIndex)) == zh_ld32(ZE_SRC(ip0))) {           // ZSTD_match4Found_cmov
                    uint32_t m = matchIndex;
                    const uint32_t offset = ip0 - m;
                    mLength = ze_common_len(ZE_SRC(ip0 + 4), ZE_SRC(m + 4), ZE_SRC(iend)) + 4;
                    while (ip0 > anchor && m > CE && ze_sp_byte(sp, ip0 - 1) == ze_sp_byte(sp, m - 1)) { ip0--; m--; mLength++; }
                    off2 = off1; off1 = offset;
                    ZE_STORE(ip0 - anchor, offset + 3, mLength);
                    break;
                }
                dEnt = dHash[dTag1 >> 8]; dictTagsMatch = (dEnt & 255) == (dTag1 & 255);
                matchIndex = hashTable[hash1];
                if (ip1 >= nextStep) { step++; nextStep += 256; }
                ip0 = ip1; ip1 = ip1 + step;
                if (ip1 > ilimit) { done = true; break; }
                curr = ip0; hash0 = hash1;
            }
            if (done) break;
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {
                hashTable[ze_hash(ZE_SRC(curr + 2), hlog, mls)] = curr + 2;
                hashTable[ze_hash(ZE_SRC(ip0 - 2), hlog, mls)] = ip0 - 2;
                while (ip0 <= ilimit) {
                    const uint32_t rep2 = ip0 - off2;
                    if (!((uint32_t)((CE - 1) - rep2) >= 3 && ze_sp_rd32(sp, rep2) == zh_ld32(ZE_SRC(ip0)))) break;
                    const uint32_t r = ze_sp_count(sp, ip0 + 4, rep2 + 4) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    ZE_STORE(0, 1, r);
                    hashTable[ze_hash(ZE_SRC(ip0), hlog, mls)] = ip0;
                    ip0 += r; anchor = ip0;
                }
            }
            ip1 = ip0 + stepSize;
        }
    }
#undef ZE_STORE
    {   const uint32_t lastLL = iend - anchor; const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = a_[i]; lp += lastLL; }
#undef ZE_SRC
    *pLit = (uint32_t)(lp - lits);
    return nseq;
}

// ------------------------------------------------------------------------------------------ searches of libzstd's table-COPY mode
// Above the attach cutoff (ZSTD_shouldAttachDict, zstd.c:25263) libzstd copies the dictionary's tables into the frame's own
// (ZSTD_resetCCtx_byCopyingCDict, zstd.c:25356: tags stripped, the dictionary's hash / chain logs kept as they are) and the dictionary
// content becomes an EXTERNAL segment of the window, searched by the _extDict variants. Same index space as above (content byte k is
// index 2 + k, the source starts at CE), ONE table set holding dictionary and source positions alike. One lane each.
// ZSTD_compressBlock_fast_extDict_generic (zstd.c:32423).
// frame + blkOff: the block (several per frame above 128 KiB: the tables and rep[] carry over, the dictionary stays in the window -- the
// caller refuses frames whose window would drop it). rep: in / out.
ZH_DEVFN uint32_t ze_fast_ext(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* frame, uint32_t blkOff, uint32_t srcSize, const ZePar& cp,
                              const ZeCDict& cd, const uint8_t* content, uint32_t* table, uint32_t* rep)
{
    const int hlog = cp.hlog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t stepSize = (uint32_t)cp.tlen + (cp.tlen == 0) + 1;
    const uint32_t CE = 2 + cd.contentSize, DS = 2;
    const uint8_t* const src = frame;
    ZeSpace sp; sp.content = content; sp.src = src; sp.CE = CE; sp.end = CE + blkOff + srcSize;
    const uint32_t iend = CE + blkOff + srcSize;
    uint32_t ip0 = CE + blkOff, anchor = ip0;
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    {   const uint32_t maxRep = ip0 - DS;
        if (off2 >= maxRep) { saved2 = off2; off2 = 0; }
        if (off1 >= maxRep) { saved1 = off1; off1 = 0; } }
    uint32_t nseq = 0; uint8_t* lp = lits;
#define ZE_SRC(i) (src + ((i) - CE))
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = a_[i_]; lp += ll_; \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
    if (srcSize >= 8) {
        const uint32_t ilimit = iend - 8;
        for (;;) {
            uint32_t step = stepSize, nextStep = ip0 + 128;              // kStepIncr = 1 << (kSearchStrength - 1)
            uint32_t ip1 = ip0 + 1, ip2 = ip0 + step, ip3 = ip2 + 1;
            if (ip3 >= ilimit) break;
            uint32_t hash0 = ze_hash(ZE_SRC(ip0), hlog, mls), hash1 = ze_hash(ZE_SRC(ip1), hlog, mls);
            uint32_t idx = table[hash0];
            uint32_t current0 = 0, offBase = 0, mLength = 0, m = 0;
            int found = 0;
            do {
                {   const uint32_t repIndex = ip2 - off1;                // repcode at ip2
                    const bool valid = ((uint32_t)(CE - repIndex) >= 4) & (off1 > 0);      /* intentional underflow */
                    const uint32_t rval = valid ? ze_sp_rd32(sp, repIndex) : zh_ld32(ZE_SRC(ip2)) ^ 1u;
                    current0 = ip0; table[hash0] = current0;
                    if (zh_ld32(ZE_SRC(ip2)) == rval) {
                        ip0 = ip2; m = repIndex;
                        mLength = ze_sp_byte(sp, ip0 - 1) == ze_sp_byte(sp, m - 1) ? 1u : 0u;
                        ip0 -= mLength; m -= mLength;
                        offBase = 1; mLength += 4;
                        found = 1; break;
                    } }
                if (idx >= DS && ze_sp_rd32(sp, idx) == zh_ld32(ZE_SRC(ip0))) { found = 2; break; }
                idx = table[hash1];
                hash0 = hash1; hash1 = ze_hash(ZE_SRC(ip2), hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip3;
                current0 = ip0; table[hash0] = current0;
                if (idx >= DS && ze_sp_rd32(sp, idx) == zh_ld32(ZE_SRC(ip0))) { found = 2; break; }
                idx = table[hash1];
                hash0 = hash1; hash1 = ze_hash(ZE_SRC(ip2), hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
                if (ip2 >= nextStep) { step++; nextStep += 128; }
            } while (ip3 < ilimit);
            if (!found) break;
            if (found == 2) {
                const uint32_t offset = current0 - idx, low = idx < CE ? DS : CE;
                m = idx;
                off2 = off1; off1 = offset; offBase = offset + 3; mLength = 4;
                while (ip0 > anchor && m > low && ze_sp_byte(sp, ip0 - 1) == ze_sp_byte(sp, m - 1)) { ip0--; m--; mLength++; }
            }
            mLength += ze_sp_count(sp, ip0 + mLength, m + mLength);
            ZE_STORE(ip0 - anchor, offBase, mLength);
            ip0 += mLength; anchor = ip0;
            if (ip1 < ip0) table[hash1] = ip1;
            if (ip0 <= ilimit) {
                table[ze_hash(ZE_SRC(current0 + 2), hlog, mls)] = current0 + 2;
                table[ze_hash(ZE_SRC(ip0 - 2), hlog, mls)] = ip0 - 2;
                while (ip0 <= ilimit) {
                    const uint32_t rep2 = ip0 - off2;
                    if (!(((uint32_t)((CE - 1) - rep2) >= 3) & (off2 > 0)) || ze_sp_rd32(sp, rep2) != zh_ld32(ZE_SRC(ip0))) break;
                    const uint32_t r = ze_sp_count(sp, ip0 + 4, rep2 + 4) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    ZE_STORE(0, 1, r);
                    table[ze_hash(ZE_SRC(ip0), hlog, mls)] = ip0;
                    ip0 += r; anchor = ip0;
                }
            }
        }
    }
#undef ZE_STORE
    {   const uint32_t lastLL = iend - anchor; const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = a_[i]; lp += lastLL; }
#undef ZE_SRC
    if (saved1 != 0 && off1 != 0) saved2 = saved1;
    rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
    *pLit = (uint32_t)(lp - lits);
    return nseq;
}
// ZSTD_compressBlock_doubleFast_extDict_generic (zstd.c:31544).
ZH_DEVFN uint32_t ze_dfast_ext(uint64_t* seqs, uint8_t* lits, uint32_t* pLit, const uint8_t* frame, uint32_t blkOff, uint32_t srcSize, const ZePar& cp,
                               const ZeCDict& cd, const uint8_t* content, uint32_t* hashLong, uint32_t* hashSmall, uint32_t* rep)
{
    const int hl = cp.hlog, hs = cp.clog;
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    const uint32_t CE = 2 + cd.contentSize, DS = 2;
    const uint8_t* const src = frame;
    ZeSpace sp; sp.content = content; sp.src = src; sp.CE = CE; sp.end = CE + blkOff + srcSize;
    const uint32_t iend = CE + blkOff + srcSize;
    uint32_t ip = CE + blkOff, anchor = ip;
    uint32_t off1 = rep[0], off2 = rep[1];
    uint32_t nseq = 0; uint8_t* lp = lits;
#define ZE_SRC(i) (src + ((i) - CE))
#define ZE_STORE(LL, OFFBASE, ML) do { const uint32_t ll_ = (uint32_t)(LL); const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i_ = 0; i_ < ll_; i_++) lp[i_] = a_[i_]; lp += ll_; \
        seqs[nseq] = ZE_SEQ_PACK(OFFBASE, ll_, (uint32_t)(ML)); nseq++; } while (0)
#define ZE_BACKX(M) do { const uint32_t low_ = (M) < CE ? DS : CE; while (ip > anchor && (M) > low_ && ze_sp_byte(sp, ip - 1) == ze_sp_byte(sp, (M) - 1)) { ip--; (M)--; mLength++; } } while (0)
    if (srcSize >= 8) {
        const uint32_t ilimit = iend - 8;
        while (ip < ilimit) {
            const uint32_t hSmall = ze_hash(ZE_SRC(ip), hs, mls), hLong = ze_hash(ZE_SRC(ip), hl, 8);
            uint32_t matchIndex = hashSmall[hSmall], matchLongIndex = hashLong[hLong];
            const uint32_t curr = ip;
            const uint32_t repIndex = curr + 1 - off1;
            uint32_t mLength = 0, offset = 0;
            hashSmall[hSmall] = curr; hashLong[hLong] = curr;
            if ((((uint32_t)((CE - 1) - repIndex) >= 3) & (off1 <= curr + 1 - DS)) && ze_sp_rd32(sp, repIndex) == zh_ld32(ZE_SRC(ip + 1))) {
                mLength = ze_sp_count(sp, ip + 1 + 4, repIndex + 4) + 4;
                ip++;
                ZE_STORE(ip - anchor, 1, mLength);
            } else {
                if (matchLongIndex > DS && ze_sp_rd64(sp, matchLongIndex) == zh_ld64(ZE_SRC(ip))) {
                    mLength = ze_sp_count(sp, ip + 8, matchLongIndex + 8) + 8;
                    offset = curr - matchLongIndex;
                    ZE_BACKX(matchLongIndex);
                } else if (matchIndex > DS && ze_sp_rd32(sp, matchIndex) == zh_ld32(ZE_SRC(ip))) {
                    const uint32_t h3 = ze_hash(ZE_SRC(ip + 1), hl, 8);
                    uint32_t matchIndex3 = hashLong[h3];
                    hashLong[h3] = curr + 1;
                    if (matchIndex3 > DS && ze_sp_rd64(sp, matchIndex3) == zh_ld64(ZE_SRC(ip + 1))) {
                        mLength = ze_sp_count(sp, ip + 9, matchIndex3 + 8) + 8;
                        ip++;
                        offset = curr + 1 - matchIndex3;
                        ZE_BACKX(matchIndex3);
                    } else {
                        mLength = ze_sp_count(sp, ip + 4, matchIndex + 4) + 4;
                        offset = curr - matchIndex;
                        ZE_BACKX(matchIndex);
                    }
                } else { ip += ((ip - anchor) >> 8) + 1; continue; }
                off2 = off1; off1 = offset;
                ZE_STORE(ip - anchor, offset + 3, mLength);
            }
            ip += mLength; anchor = ip;
            if (ip <= ilimit) {
                const uint32_t ins = curr + 2;
                hashLong[ze_hash(ZE_SRC(ins), hl, 8)] = ins;
                hashLong[ze_hash(ZE_SRC(ip - 2), hl, 8)] = ip - 2;
                hashSmall[ze_hash(ZE_SRC(ins), hs, mls)] = ins;
                hashSmall[ze_hash(ZE_SRC(ip - 1), hs, mls)] = ip - 1;
                while (ip <= ilimit) {
                    const uint32_t rep2 = ip - off2;
                    if (!((((uint32_t)((CE - 1) - rep2) >= 3) & (off2 <= ip - DS)) && ze_sp_rd32(sp, rep2) == zh_ld32(ZE_SRC(ip)))) break;
                    const uint32_t r = ze_sp_count(sp, ip + 4, rep2 + 4) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    ZE_STORE(0, 1, r);
                    hashSmall[ze_hash(ZE_SRC(ip), hs, mls)] = ip;
                    hashLong[ze_hash(ZE_SRC(ip), hl, 8)] = ip;
                    ip += r; anchor = ip;
                }
            }
        }
    }
#undef ZE_BACKX
#undef ZE_STORE
    {   const uint32_t lastLL = iend - anchor; const uint8_t* a_ = ZE_SRC(anchor); for (uint32_t i = 0; i < lastLL; i++) lp[i] = a_[i]; lp += lastLL; }
#undef ZE_SRC
    rep[0] = off1; rep[1] = off2;
    *pLit = (uint32_t)(lp - lits);
    return nseq;
}

// working parameters of a frame compressed against an attached dictionary: the dictionary's own row shrunk to the source
// (ZSTD_resetCCtx_byAttachingCDict, zstd.c:25279); the window log stays the one chosen for the source.
ZH_DEV void ze_dict_cparams(ZePar& cp, const ZeCDict& cd, uint32_t srcSize)
{
    int w = 31, h = cd.hlog, c = cd.clog;
    const int srcLog = srcSize < 64 ? 6 : zh_highbit32(srcSize - 1) + 1;
    if (w > srcLog) w = srcLog;
    if (h > w + 1) h = w + 1;
    if (c > w) c = w;
    cp.hlog = h; cp.clog = c; cp.mml = cd.mml; cp.strat = cd.strat; cp.tlen = cd.tlen;
}
ZH_DEV uint32_t ze_dict_attach_max(const ZeCDict& cd) { return cd.strat == 1 ? ZE_DICT_ATTACH_MAX_FAST : ZE_DICT_ATTACH_MAX; }
// what the match kernels of a dictionary batch take: sources up to the attach cutoff -- or up to what their table / arena slots were sized
// for, where the caller's size hint made them smaller (262 144 slots of 48 KiB instead of 192 KiB: configs[3] in ONE launch); the rest is the generic kernel's
ZH_DEV uint32_t ze_dict_slot_max(const ZhipEncodeArgs& a) { const uint32_t m = ze_dict_attach_max(*a.cdict); return a.slotSrcMax && a.slotSrcMax < m ? a.slotSrcMax : m; }
// working parameters of libzstd's table-copy mode (ZSTD_resetCCtx_byCopyingCDict, zstd.c:25368-25373): everything from the dictionary's
// row UNCHANGED, the window log as ZSTD_getCParamsFromCCtxParams chooses it for source + dictionary content (row of that total size,
// clamped to its log2). 0, or parameter_unsupported when the window would drop the dictionary before the frame ends.
ZH_DEV int ze_dict_copy_cparams(ZePar& cp, const ZeCDict& cd, const ZeRows& rows, uint32_t srcSize)
{
    ZePar t;
    const uint32_t total = srcSize + cd.contentSize;
    const uint32_t tableID = (total <= 256u * 1024) + (total <= 128u * 1024) + (total <= 16u * 1024);
    int w = rows.r[tableID][0];
    const int srcLog = total < 64 ? 6 : zh_highbit32(total - 1) + 1;
    if (w > srcLog) w = srcLog;
    if (w < 10) w = 10;
    (void)t;
    cp.wlog = w; cp.hlog = cd.hlog; cp.clog = cd.clog; cp.mml = cd.mml; cp.strat = cd.strat; cp.tlen = cd.tlen;
    // the dictionary stays valid while the SOURCE fits the window (ZSTD_checkDictValidity, zstd.c:19360: block end > dictionary end + window);
    // beyond that libzstd drops it part-way through the frame, which is not implemented: refused
    if (w > 27 || ((uint64_t)1 << w) < (uint64_t)srcSize) return ZE_PARAM_UNSUPPORTED;
    return 0;
}

// ------------------------------------------------------------------------------------------ sequences section
// ZSTD_selectEncodingType (zstd.c:21252), strategy below "lazy", first block: 0 basic, 1 rle, 2 compressed
ZH_DEV int ze_select_mode(uint32_t mostFrequent, uint32_t nbSeq, uint32_t defLog, bool defaultAllowed, uint32_t repeatMode, uint32_t strat)
{
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        const uint32_t dynMin = ((1u << defLog) * (10u - strat)) >> 3;      // mult = 10 - strategy (fast 1, double-fast 2)
        if (repeatMode == 2 && nbSeq < 1000) return 3;          // set_repeat: the dictionary's table
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defLog - 1))) return 0;
    }
    return 2;
}
// ZSTD_buildCTable (zstd.c:21338) for one of LL / OF / ML, by the whole wave (all lanes call; results are wave-uniform). Returns the
// header bytes written at `out`.
ZH_DEVFN uint32_t ze_build_seq_table(ZeLDS& L, int which, uint8_t* out, int* mode, uint32_t firstCode, uint32_t lastCode, uint32_t nbSeq, const ZeCDict* cd, const uint32_t* fseRep, uint32_t strat)
{
    const uint32_t lane = zh_lane();
    const uint32_t maxCode = which == 0 ? 35 : which == 1 ? 31 : 52;
    const uint32_t fseLog = which == 1 ? 8 : 9, defLog = which == 1 ? 5 : 6, defMax = which == 0 ? 35 : which == 1 ? 28 : 52;
    const int16_t* defNorm = which == 0 ? ze_llDef : which == 1 ? ze_ofDef : ze_mlDef;
    uint32_t* count = L.cnt[which];
    int16_t* const norm = ze_norm_area(L);
    const uint32_t c = lane <= maxCode ? count[lane] : 0u;
    const uint64_t present = zh_ballot(c != 0);
    const uint32_t max = present ? 63u - (uint32_t)zh_clz64(present) : 0u, most = zh_wave_max(c);
    const bool defaultAllowed = which != 1 || max <= 28;
    const uint32_t repeatMode = fseRep ? fseRep[which] : !cd ? 0u : which == 0 ? cd->llRepeat : which == 1 ? cd->ofRepeat : cd->mlRepeat;   // (fseRep: a later block's view)
    *mode = ze_select_mode(most, nbSeq, defLog, defaultAllowed, repeatMode, strat);
    ZeCTab& t = L.tab[which];
    if (*mode == 3) return 0;        // set_repeat: the dictionary's table, copied into LDS by the whole wave (ze_copy_dict_tables)
    if (*mode == 1) {
        if (zh_opaque(lane) == 0) { ze_fse_build_rle(t, firstCode); out[0] = (uint8_t)firstCode; }
        ze_fence(); zh_sync();
        return 1;
    }
    if (*mode == 0) {
        if (lane <= defMax) norm[lane] = defNorm[lane];
        ze_fence(); zh_sync();
        ze_fse_build_ctab_wave(t, norm, defMax, defLog, ze_cell_sym(L), (uint16_t*)L.stack, ze_fill_area(L));
        return 0;
    }
    const uint32_t lg = ze_fse_optimal_log(fseLog, nbSeq, max, 2);
    uint32_t n1 = nbSeq;
    if (zh_first(count[lastCode]) > 1) {                                          // the last sequence's symbols are never decoded FROM: one count less
        zh_sync();
        if (zh_opaque(lane) == 0) count[lastCode]--;
        n1--;
        ze_fence(); zh_sync();
    }
    ze_fse_normalize_wave(norm, lg, count, n1, max, n1 >= 2048);
    ze_fence(); zh_sync();
    const uint32_t h = ze_fse_write_ncount_wave(out, norm, max, lg, (uint32_t*)((uint8_t*)L.node + 2048));
    ze_fse_build_ctab_wave(t, norm, max, lg, ze_cell_sym(L), (uint16_t*)L.stack, ze_fill_area(L));
    return h;
}
// the tables whose mode is set_repeat (3): dictionary -> LDS, 4 bytes per lane per step, all loads of a table in flight together
ZH_DEV void ze_copy_dict_tables(ZeLDS& L, const ZeCDict* cd, uint32_t modes /* mLL | mOF << 2 | mML << 4 */)
{
    const uint32_t lane = zh_lane();
    constexpr uint32_t W = sizeof(ZeCTab) / 4;
    static_assert(sizeof(ZeCTab) % 4 == 0, "ZeCTab is copied as dwords");
    for (uint32_t which = 0; which < 3; which++) {
        if (((modes >> (2 * which)) & 3) != 3) continue;
        const uint32_t* src = (const uint32_t*)&cd->tab[which];
        uint32_t* dst = (uint32_t*)&L.tab[which];
        uint32_t r[(W + 63) / 64];
#pragma unroll
        for (uint32_t q = 0; q < (W + 63) / 64; q++) r[q] = lane + 64 * q < W ? src[lane + 64 * q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < (W + 63) / 64; q++) if (lane + 64 * q < W) dst[lane + 64 * q] = r[q];
    }
}
// ZSTD_LLcode / ZSTD_MLcode (zstd.c:19738, :19755) and the extra-bit counts (LL_bits / ML_bits), computed: the reference's lookup
// tables would be per-lane reads of global memory in the middle of per-sequence work. Nibble k of 0x5555444433221100 is the
// code step inside [16, 32) of the LL scale / [32, 48) of the ML scale: 0 0 1 1 2 2 3 3 4 4 4 4 5 5 5 5.
ZH_DEV uint32_t ze_ll_code(uint32_t v)
{
    if (v < 16) return v;
    if (v < 32) return 16 + (uint32_t)((0x5555444433221100ull >> (4 * (v - 16))) & 15);
    if (v < 64) { const uint32_t t = (v - 32) >> 3; return 22 + (t > 2 ? 2u : t); }
    return (uint32_t)zh_highbit32(v) + 19;
}
ZH_DEV uint32_t ze_ml_code(uint32_t ml)
{
    const uint32_t b = ml - 3;
    if (b < 32) return b;
    if (b < 48) return 32 + (uint32_t)((0x5555444433221100ull >> (4 * (b - 32))) & 15);
    if (b < 64) return 38 + ((b - 48) >> 3);
    if (b < 128) { const uint32_t t = (b - 64) >> 4; return 40 + (t > 2 ? 2u : t); }
    return (uint32_t)zh_highbit32(b) + 36;
}
ZH_DEV uint32_t ze_ll_bits(uint32_t c) { return c < 16 ? 0u : c < 25 ? (uint32_t)((0x433221111ull >> (4 * (c - 16))) & 15) : c - 19; }
ZH_DEV uint32_t ze_ml_bits(uint32_t c) { return c < 32 ? 0u : c < 43 ? (uint32_t)((0x54433221111ull >> (4 * (c - 32))) & 15) : c - 36; }

// ------------------------------------------------------------------------------------------ sequences bitstream, wave-parallel
#ifndef ZE_CHAIN_UNROLL
#define ZE_CHAIN_UNROLL 1
#endif
// ZSTD_encodeSequences_body (zstd.c:21386): sequences last to first; per sequence the OF, ML, LL state transitions
// (FSE_encodeSymbol :2785) then the LL, ML, OF extra bits; finally the three states and the end mark. The only serial part is the
// three state chains -- one lane per table walks them, 64 sequences per round, through LDS -- while what each sequence contributes
// to the stream (up to 89 bits) is assembled by its own lane, placed by a wave prefix sum of the bit counts and ORed into an LDS
// bit buffer that is flushed to the frame in whole bytes (the odd bits carry into the next round). All lanes call.
// Returns the stream's size, 0 when it does not fit in cap.
ZH_COLD uint32_t ze_encode_sequences_wave(ZeLDS& L, uint8_t* out, uint32_t cap, const uint64_t* seqs, uint32_t nbSeq, ZeProf* P = nullptr)
{
#ifdef ZE_PROF_STREAM
    uint64_t sq0 = P ? zd_clock() : 0;
#define ZE_SQT(i) do { if (P) { const uint64_t t_ = zd_clock(); P->acc[i] += t_ - sq0; sq0 = t_; } } while (0)
#else
#define ZE_SQT(i) do { } while (0)
#endif
    const uint32_t lane = zh_lane();
    uint32_t* const tt = ze_scratch(L);          // per symbol: deltaNbBits, deltaFindState (FSE_symbolCompressionTransform); LL at 0, OF at 36, ML at 68
    uint32_t* const pre = tt + 2 * 121;          // [table][slot] -> the slot's sequence's two constants (3 x 64 x 2)
    uint32_t* const rec = pre + 384;             // [table][slot]: state | nbBits << 16
    uint32_t* const bitbuf = rec + 192;          // 196 dwords
    static_assert((2 * 121 + 384 + 192 + 196) * 4 <= sizeof(L.node), "sequence encoder scratch must fit the tree-node area");
    zh_sync();
    for (uint32_t k = lane; k < 192; k += 64) {
        const uint32_t t = k >> 6, sy = k & 63;
        const ZeCTab& T = L.tab[t];
        const uint32_t size = t == 0 ? 36u : t == 1 ? 32u : 53u, base = t == 0 ? 0u : t == 1 ? 36u : 68u;
        uint32_t dnb = 0, dfs = 0;
        if (T.log != 0 && sy <= T.maxSym && T.norm[sy] != 0) {
            const uint32_t c = T.norm[sy] == -1 ? 1u : (uint32_t)T.norm[sy];
            const uint32_t mb = c > 1 ? (uint32_t)T.log - (uint32_t)zh_highbit32(c - 1) : (uint32_t)T.log;
            dnb = (mb << 16) - (c << mb);
            dfs = (uint32_t)T.cellOf[sy] - c;
        }
        if (sy < size) { tt[2 * (base + sy)] = dnb; tt[2 * (base + sy) + 1] = dfs; }
    }
    for (uint32_t k = lane; k < 196; k += 64) bitbuf[k] = 0;
    zh_sync();
    uint32_t v = 0;                               // lanes 0..2: the LL / OF / ML state
    uint32_t carryBits = 0, bytePos = 0; bool ovf = false;
    uint32_t hi = nbSeq - 1;
    for (;;) {
        const bool valid = lane <= hi;
        const uint64_t q = valid ? seqs[hi - lane] : 0;
        const uint32_t ll = ZE_SEQ_LL(q), ml = ZE_SEQ_ML(q), ob = ZE_SEQ_OFF(q);
        uint32_t lc = 0, oc = 0, mc = 0;
        if (valid) { lc = ze_ll_code(ll); oc = (uint32_t)zh_highbit32(ob); mc = ze_ml_code(ml); }
        // every lane looks up its own sequence's transition constants for the three tables, so the chain lanes below only do the
        // state-dependent part: two adds, two shifts, one table read per step
        {   const uint32_t* eL = tt + 2 * lc; const uint32_t* eO = tt + 2 * (36 + oc); const uint32_t* eM = tt + 2 * (68 + mc);
            pre[2 * lane] = eL[0]; pre[2 * lane + 1] = eL[1];
            pre[128 + 2 * lane] = eO[0]; pre[128 + 2 * lane + 1] = eO[1];
            pre[256 + 2 * lane] = eM[0]; pre[256 + 2 * lane + 1] = eM[1]; }
        zh_sync();
        ZE_SQT(ZEP_SQ_PRE);
        if (zh_opaque(lane) < 3) {
            const uint32_t t = lane;
            const ZeCTab& T = L.tab[t];
            const uint32_t cnt = hi < 63 ? hi + 1 : 64;
            const uint32_t* pt = pre + 128 * t;
            uint32_t* rt = rec + 64 * t;
            if (T.log == 0) { for (uint32_t j = 0; j < cnt; j++) rt[j] = 0; v = 0; }
            else {
                uint32_t j = 0;
                if (hi == nbSeq - 1) {                                      // FSE_initCState2, zstd.c:2774
                    const uint32_t dnb = pt[0], dfs = pt[1];
                    const uint32_t nbOut = (dnb + (1u << 15)) >> 16;
                    const uint32_t value = (nbOut << 16) - dnb;
                    v = T.next[(value >> nbOut) + dfs];
                    rt[0] = 0; j = 1;
                }
                // the chain: add, shift, shift, add, one table read per step; the next slot's two constants are requested right
                // behind that read (LDS answers in order), so their latency is not on the chain
                uint64_t c = *(const uint64_t*)(pt + 2 * (j < cnt ? j : 0));
#if ZE_CHAIN_UNROLL
                // (a full round -- every round but a frame's first and last -- in steps of eight with constant offsets: the loop's counter, bound check and
                // address arithmetic are 4 of its 15 instructions, and three busy lanes pay for every one of them)
                if (cnt == 64 && j == 0) {
                    const auto nx = ZH_LDS_CPTR(uint16_t, T.next);
#pragma nounroll
                    for (uint32_t j4 = 0; j4 < 64; j4 += 8) {                  // (eight steps per trip and no further: the whole round unrolled is 733 instructions that 4 KiB documents run three times each -- E2 4.8 -> 5.6 ms per dictionary batch, r06x)
                        const uint32_t* p4 = pt + 2 * j4; uint32_t* r4 = rt + j4;
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) {
                            const uint32_t dnb = (uint32_t)c, dfs = (uint32_t)(c >> 32);
                            const uint32_t nb = (v + dnb) >> 16;
                            r4[u] = v | (nb << 16);
                            const uint32_t vn = nx[(v >> nb) + dfs];
                            c = *(const uint64_t*)(p4 + 2 * (u + 1 < 8 || j4 + 8 < 64 ? u + 1 : 0));
                            v = vn;
                        }
                    }
                    j = 64;
                }
#endif
                while (j < cnt) {
                    const uint32_t dnb = (uint32_t)c, dfs = (uint32_t)(c >> 32);
                    const uint32_t nb = (v + dnb) >> 16;
                    rt[j] = v | (nb << 16);                                 // the state before the transition; its low nb bits go to the stream
                    const uint32_t vn = T.next[(v >> nb) + dfs];
                    j++;
                    c = *(const uint64_t*)(pt + 2 * (j < cnt ? j : 0));
                    v = vn;
                }
            }
        }
        zh_sync();
        ZE_SQT(ZEP_SQ_CHAIN);
        uint64_t lo = 0, up = 0; uint32_t pos = 0;
#define ZE_ADD(val, nbits) do { const uint32_t n_ = (nbits); const uint64_t v_ = (val); if (n_) { if (pos < 64) { lo |= v_ << pos; if (pos + n_ > 64) up |= v_ >> (64 - pos); } \
                                else up |= v_ << (pos - 64); pos += n_; } } while (0)
        if (valid) {
            const uint32_t rL = rec[lane], rO = rec[64 + lane], rM = rec[128 + lane];
            ZE_ADD(rO & ((1u << (rO >> 16)) - 1), rO >> 16); ZE_ADD(rM & ((1u << (rM >> 16)) - 1), rM >> 16); ZE_ADD(rL & ((1u << (rL >> 16)) - 1), rL >> 16);
            const uint32_t lb = ze_ll_bits(lc), mb = ze_ml_bits(mc);
            ZE_ADD(ll & ((1u << lb) - 1), lb); ZE_ADD((ml - 3) & ((1u << mb) - 1), mb); ZE_ADD(ob & ((1u << oc) - 1), oc);
        }
#undef ZE_ADD
        const uint32_t incl = zh_scan_add(pos);
        const uint32_t tot = zh_shfl(incl, 63);
        if (pos) {
            const uint32_t P = carryBits + incl - pos, sh = P & 31;
            uint32_t* w = bitbuf + (P >> 5);
            const uint64_t A = lo << sh, B = (up << sh) | (sh ? lo >> (64 - sh) : 0ull);
            if ((uint32_t)A) zh_lds_atomic_or(w, (uint32_t)A);
            if ((uint32_t)(A >> 32)) zh_lds_atomic_or(w + 1, (uint32_t)(A >> 32));
            if ((uint32_t)B) zh_lds_atomic_or(w + 2, (uint32_t)B);
            if ((uint32_t)(B >> 32)) zh_lds_atomic_or(w + 3, (uint32_t)(B >> 32));
        }
        zh_sync();
        ZE_SQT(ZEP_SQ_PACK);
        const uint32_t totalBits = carryBits + tot, nbytes = totalBits >> 3;
        if (bytePos + nbytes > cap) ovf = true;
        if (!ovf) {
            const uint32_t nd = nbytes >> 2;
            for (uint32_t i = lane; i < nd; i += 64) zh_st32(out + bytePos + 4 * i, bitbuf[i]);
            if (lane < (nbytes & 3)) out[bytePos + 4 * nd + lane] = ((const uint8_t*)bitbuf)[4 * nd + lane];
        }
        const uint32_t rem = totalBits & 7;
        const uint32_t cb = rem ? ((const uint8_t*)bitbuf)[nbytes] : 0u;
        zh_sync();
        for (uint32_t k = lane; k < 196; k += 64) bitbuf[k] = k == 0 ? cb : 0u;
        zh_sync();
        carryBits = rem; bytePos += nbytes;
        ZE_SQT(ZEP_SQ_FLUSH);
        if (hi < 64) break;
        hi -= 64;
    }
#undef ZE_SQT
    if (zh_opaque(lane) < 3) L.misc[12 + lane] = v;
    zh_sync();
    if (zh_opaque(lane) == 0) {
        uint64_t acc = bitbuf[0] & 0xFFu; uint32_t nb = carryBits;
        const uint32_t lgL = (uint32_t)L.tab[0].log, lgO = (uint32_t)L.tab[1].log, lgM = (uint32_t)L.tab[2].log;
        acc |= (uint64_t)(L.misc[14] & ((1u << lgM) - 1)) << nb; nb += lgM;
        acc |= (uint64_t)(L.misc[13] & ((1u << lgO) - 1)) << nb; nb += lgO;
        acc |= (uint64_t)(L.misc[12] & ((1u << lgL) - 1)) << nb; nb += lgL;
        acc |= 1ull << nb; nb++;
        const uint32_t nby = (nb + 7) >> 3;
        if (ovf || bytePos + nby > cap) L.misc[11] = 0;
        else { for (uint32_t i = 0; i < nby; i++) out[bytePos + i] = (uint8_t)(acc >> (8 * i)); L.misc[11] = bytePos + nby; }
    }
    ze_fence();
    zh_sync();
    const uint32_t r = zh_first(L.misc[11]);
    zh_sync();
    return r;
}

// The literals of a block from its sequence list: every source byte the matches do not cover, in order (what ZSTD_storeSeq's
// literal copy accumulates, zstd.c:19930). The match-finding kernel records sequences only; this runs wave-parallel: 64 sequences
// per round, two prefix sums place the literal runs, each lane copies its own run when it is short, the wave copies the long
// ones together. All lanes call. Returns the literal count.
ZH_COLD uint32_t ze_gather_literals(ZeLDS& L, uint8_t* lits, const uint8_t* src, uint32_t srcSize, const uint64_t* seqs, uint32_t nbSeq)
{
    const uint32_t lane = zh_lane();
    uint32_t* const uend = ze_scratch(L);                                   // per round: inclusive 16-byte-unit ends of the long runs
    uint32_t* const rsrc = uend + 64; uint32_t* const rdst = rsrc + 64; uint32_t* const rlen = rdst + 64;
    uint32_t litBase = 0, srcBase = 0;
    uint64_t qn = lane < nbSeq ? seqs[lane] : 0;                            // the next round's sequences are requested a round ahead
    for (uint32_t b0 = 0; b0 < nbSeq; b0 += 64) {
        const uint64_t q = qn;
        qn = b0 + 64 + lane < nbSeq ? seqs[b0 + 64 + lane] : 0;
        const uint32_t ll = ZE_SEQ_LL(q), ml = ZE_SEQ_ML(q);
        const uint32_t le = zh_scan_add(ll), se = zh_scan_add(ll + ml);
        const uint32_t T = zh_shfl(le, 63), S = zh_shfl(se, 63);
        const uint32_t myLit = litBase + le - ll, mySrc = srcBase + se - ll - ml;
        // runs of up to 16 bytes (most of them): the run's owner loads 16 source bytes in one go (the load may reach past the run,
        // never past the source) and stores exactly its ll bytes. Longer runs are cut into 16-byte units -- the last one shifted back
        // to end with the run -- and ALL the round's units are spread over the lanes (a unit finds its run by a binary search over
        // the unit prefix sums), so a round costs one memory round trip whatever its run lengths are.
        const bool small = ll > 0 && ll <= 16 && mySrc + 16 <= srcSize;
        const bool tiny = ll > 0 && ll < 16 && !small;                      // a short run within 16 bytes of the source's end
        const uint32_t units = ll > 16 ? (ll + 15) >> 4 : 0u;
        const uint32_t ue = zh_scan_add(units);
        const uint32_t U = zh_shfl(ue, 63);
        zh_v16 v; v.lo = 0; v.hi = 0;
        if (small) v = zh_ld128(src + mySrc);
        if (U) {
            zh_sync();
            uend[lane] = ue; rsrc[lane] = mySrc; rdst[lane] = myLit; rlen[lane] = ll;
            zh_sync();
            for (uint32_t u = lane; u < U; u += 64) {
                uint32_t j = 0;                                                // smallest j with uend[j] > u
                for (uint32_t stp = 32; stp; stp >>= 1) if (uend[j + stp - 1] <= u) j += stp;
                const uint32_t k = u - (j ? uend[j - 1] : 0u), len = rlen[j];
                const uint32_t off = 16 * k + 16 <= len ? 16 * k : len - 16;
                *(zh_v16*)(lits + rdst[j] + off) = zh_ld128(src + rsrc[j] + off);
            }
        }
        if (small) {
            uint8_t* d = lits + myLit;
            uint64_t w = v.lo;
            for (uint32_t k = 0; k < 8; k++) { if (k < ll) d[k] = (uint8_t)w; w >>= 8; }
            w = v.hi;
            for (uint32_t k = 8; k < 16; k++) { if (k < ll) d[k] = (uint8_t)w; w >>= 8; }
        }
        if (tiny) for (uint32_t k = 0; k < ll; k++) lits[myLit + k] = src[mySrc + k];
        litBase += T; srcBase += S;
    }
    const uint32_t tail = srcSize - srcBase;
    ze_copy_wave(lits + litBase, src + srcBase, tail);
    ze_fence();
    zh_sync();
    return litBase + tail;
}

// ZSTD_compressBlock_internal (zstd.c:27337). All lanes call. Returns body size, 0 = store raw, 1 = RLE block (out[0] = the byte).
// mb == null: the block is the whole frame. mb != null: one block of a multi-block frame -- the hash tables, the two repcodes
// (L.mrep) and the previous block's Huffman table (L.prev*) carry over and advance only when the block is emitted compressed.
struct ZePre { const uint64_t* seqs; const uint8_t* lits; uint32_t nbSeq, litSize; };   // output of the match-finding kernel
struct ZeMulti { const uint8_t* frame; bool firstBlock; ZeLDSMulti* st; uint32_t blkOff; };

// SEARCH = false (the entropy kernel, whose sequences always come from a match kernel): the search is not even compiled in -- with it, the
// 128-register entropy kernel spills in its hot loops (r02g: +350 bytes of scratch per lane made the kernel 30x slower on 4 KiB inputs)
template <bool SEARCH>
ZH_DEVFN uint32_t ze_compress_block(ZeLDS& L, uint8_t* out, uint32_t cap, const uint8_t* src, uint32_t srcSize, const ZePar& cp, uint8_t* ws,
                                    const ZePre* pre, const ZhipEncodeArgs& a, const ZeMulti* mb = nullptr, ZeProf* P = nullptr)
{
    const ZeCDict* cd = a.cdict;                                      // (multi-block frames against a dictionary: its state lives in mb->st from the second block on)
    const uint32_t* const fseRep = mb && cd ? mb->st->fseRep : nullptr;
    const uint32_t lane = zh_lane();
    if (srcSize < 7) return 0;
    uint32_t* hashLong = (uint32_t*)(ws + ZE_WS_HASHL);
    uint32_t* hashSmall = (uint32_t*)(ws + ZE_WS_HASHS);
    const uint64_t* seqs = pre ? pre->seqs : (const uint64_t*)(ws + ZE_WS_SEQ);
    const uint8_t* lits = pre && pre->lits ? pre->lits : ws + ZE_WS_LIT;
    uint32_t nbSeq, litSize;
    if (pre) {
        nbSeq = pre->nbSeq; litSize = pre->litSize;
        if (!pre->lits) litSize = ze_gather_literals(L, ws + ZE_WS_LIT, src, srcSize, seqs, nbSeq);    // sequences-only search output
        ZE_T(P, ZEP_GATHER);
    }
    else if constexpr (!SEARCH) return 0;                              // (never reached: frames without pre-computed sequences are raw or errors)
    else {
    const bool copyMode = cd && cd->contentSize && (mb || srcSize > ze_dict_attach_max(*cd));
    if (copyMode) {
        // ZSTD_copyCDictTableIntoCCtx (zstd.c:25340): the dictionary's tagged cells (index << 8 | tag) become plain indices
        if (!mb || mb->firstBlock) {
            for (uint32_t i = lane; i < (1u << cp.hlog); i += 64) hashLong[i] = a.cdictHashLong[i] >> 8;
            if (cp.strat == 2) for (uint32_t i = lane; i < (1u << cp.clog); i += 64) hashSmall[i] = a.cdictHashSmall[i] >> 8;
        }
    } else
    // fresh tables: the wave zeroes them with coalesced 8-byte stores
    if (!mb || mb->firstBlock) {
        uint64_t* a = (uint64_t*)hashLong; const uint32_t na = (1u << cp.hlog) / 2;
        for (uint32_t i = lane; i < na; i += 64) a[i] = 0;
        uint64_t* b = (uint64_t*)hashSmall; const uint32_t nb = (1u << cp.clog) / 2;
        for (uint32_t i = lane; i < nb; i += 64) b[i] = 0;
    }
    ze_fence();
    zh_sync();
    if (zh_opaque(lane) == 0) {
        uint32_t ls = 0;
        uint32_t nrep[2] = {1, 4};
        if (mb) { nrep[0] = mb->st->mrep[0]; nrep[1] = mb->st->mrep[1]; }
        else if (cd) { nrep[0] = cd->rep[0]; nrep[1] = cd->rep[1]; }
        const uint8_t* const base = mb ? mb->frame : src; const uint32_t off = mb ? mb->blkOff : 0u;
        const uint32_t ns = copyMode ? (cp.strat == 1 ? ze_fast_ext((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, base, off, srcSize, cp, *cd, a.cdictContent, hashLong, nrep)
                                                      : ze_dfast_ext((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, base, off, srcSize, cp, *cd, a.cdictContent, hashLong, hashSmall, nrep))
                          : mb ? (cp.strat == 1 ? ze_fast_g((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, mb->frame, src, srcSize, cp, hashLong, nrep)
                                                : ze_dfast_g((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, mb->frame, src, srcSize, cp, hashLong, hashSmall, nrep))
                          : (cd && cd->contentSize) ? (cp.strat == 1 ? ze_fast_dict((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, src, srcSize, cp, *cd, a.cdictContent,
                                                                                     a.cdictHashLong, hashLong)
                                                                    : ze_dfast_dict((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, src, srcSize, cp, *cd, a.cdictContent,
                                                                                      a.cdictHashLong, a.cdictHashSmall, hashLong, hashSmall))
                               : cp.strat == 1 ? ze_fast((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, src, srcSize, cp, hashLong)
                               : ze_dfast((uint64_t*)(ws + ZE_WS_SEQ), ws + ZE_WS_LIT, &ls, src, srcSize, cp, hashLong, hashSmall);
        L.misc[1] = ns; L.misc[2] = ls; L.misc[5] = nrep[0]; L.misc[6] = nrep[1];
    }
    ze_fence();
    zh_sync();
    nbSeq = zh_first(L.misc[1]); litSize = zh_first(L.misc[2]);
    zh_sync();
    }
    uint32_t pos, newMaxSym = 0xFFFFFFFFu;
    const uint32_t nextRep0 = mb ? zh_first(L.misc[5]) : 0, nextRep1 = mb ? zh_first(L.misc[6]) : 0;
    ZePrevHuf ph; const ZePrevHuf* php = nullptr;
    if (!mb && cd && cd->hufRepeat) { ph.bits = cd->hufBits; ph.code = cd->hufCode; ph.maxSym = cd->hufMaxSym; ph.repeat = cd->hufRepeat; php = &ph; }
    else if (mb && mb->st->prevRepeat) { ph.bits = mb->st->prevBits; ph.code = mb->st->prevCode; ph.maxSym = mb->st->prevMaxSym; ph.repeat = mb->st->prevRepeat; php = &ph; }
    if (cp.strat == 1 && cp.tlen > 0) {          // negative levels keep literals raw (ZSTD_literalsCompressionIsDisabled, zstd.c:24208)
        zh_sync();
        pos = ze_plain_literals(out, lits, litSize, 0u, false);
        ze_fence();
        zh_sync();
    } else pos = ze_compress_literals(L, out, cap, lits, litSize, nbSeq, php, &newMaxSym, P);
    ZE_T(P, ZEP_LITSTAT);
    // symbol histograms, wave-parallel (ZSTD_seqToCodes zstd.c:25647, HIST_countFast); the codes themselves are recomputed by the
    // stream encoder, only the first and the last sequence's are kept (rle mode / FSE_normalizeCount's last-symbol rule)
    zh_sync();
    for (uint32_t i = lane; i < 192; i += 64) (&L.cnt[0][0])[i] = 0;
    zh_sync();
    for (uint32_t i = lane; i < nbSeq; i += 64) {
        const uint64_t q = seqs[i];
        const uint32_t a = ze_ll_code(ZE_SEQ_LL(q)), o = (uint32_t)zh_highbit32(ZE_SEQ_OFF(q)), m = ze_ml_code(ZE_SEQ_ML(q));
        zh_lds_atomic_inc(&L.cnt[0][a]); zh_lds_atomic_inc(&L.cnt[1][o]); zh_lds_atomic_inc(&L.cnt[2][m]);
        if (i == 0) L.misc[8] = a | (o << 8) | (m << 16);
        if (i == nbSeq - 1) L.misc[9] = a | (o << 8) | (m << 16);
    }
    ze_fence();
    zh_sync();
    ZE_T(P, ZEP_SEQSTAT);
    uint32_t seqStart, lastCount = 0, seqModes = 0;
    {   // sequences header, then the three table descriptions, each built by the whole wave (r01: lane 0 built them, 17 % of the kernel)
        const uint32_t hb = nbSeq < 128 ? 1u : nbSeq < 0x7F00 ? 2u : 3u;
        if (zh_opaque(lane) == 0) {
            uint8_t* op = out + pos;
            if (nbSeq < 128) op[0] = (uint8_t)nbSeq;
            else if (nbSeq < 0x7F00) { op[0] = (uint8_t)((nbSeq >> 8) + 0x80); op[1] = (uint8_t)nbSeq; }
            else { op[0] = 0xFF; zh_st16(op + 1, (uint16_t)(nbSeq - 0x7F00)); }
        }
        uint32_t at = pos + hb;
        if (nbSeq) {
            const uint32_t head = at++;
            int md[3]; 
            const uint32_t c0 = zh_first(L.misc[8]), c1 = zh_first(L.misc[9]);
            for (int w = 0; w < 3; w++) {
                const uint32_t h = ze_build_seq_table(L, w, out + at, &md[w], (c0 >> (8 * w)) & 255, (c1 >> (8 * w)) & 255, nbSeq, cd, fseRep, (uint32_t)cp.strat);
                if (md[w] == 2) lastCount = h;
                at += h;
                ze_fence(); zh_sync();
            }
            if (zh_opaque(lane) == 0) out[head] = (uint8_t)((md[0] << 6) + (md[1] << 4) + (md[2] << 2));
            seqModes = (uint32_t)md[0] | ((uint32_t)md[1] << 2) | ((uint32_t)md[2] << 4);
        }
        seqStart = at;
    }
    ze_fence();
    zh_sync();
    zh_sync();
    if (cd && seqModes) { ze_copy_dict_tables(L, cd, seqModes); ze_fence(); zh_sync(); }
    ZE_T(P, ZEP_SEQTAB);
    uint32_t r = 1, cSize = seqStart;
    if (nbSeq) {
#ifdef ZE_PROF_STREAM
        const uint32_t bs = ze_encode_sequences_wave(L, out + seqStart, cap - seqStart, seqs, nbSeq, P);
#else
        const uint32_t bs = ze_encode_sequences_wave(L, out + seqStart, cap - seqStart, seqs, nbSeq);
#endif
        ZE_T(P, ZEP_SEQENC);
        if (bs == 0) r = 0;
        if (lastCount && lastCount + bs < 4) r = 0;
        cSize += bs;
    }
    if (cSize >= srcSize - ((srcSize >> 6) + 2)) r = 0;                    // ZSTD_minGain, zstd.c:19831
    r = r ? cSize : 0;
    if (mb) {
        if (!mb->firstBlock && r < 25) {          // a later block made of one byte value becomes an RLE block (zstd.c:27373-27384)
            const uint32_t b0 = src[0];
            bool diff = false;
            for (uint32_t i = lane; i < srcSize; i += 64) diff |= src[i] != b0;
            if (!zh_ballot(diff)) { if (zh_opaque(lane) == 0) out[0] = (uint8_t)b0; r = 1; }
        }
        zh_sync();
        if (r > 1) {                               // confirm: repcodes, and the Huffman table if this block carried a new one
            if (zh_opaque(lane) == 0) { mb->st->mrep[0] = nextRep0; mb->st->mrep[1] = nextRep1; }
            if (newMaxSym != 0xFFFFFFFFu) {
                for (uint32_t i = lane; i < 256; i += 64) { mb->st->prevBits[i] = L.hufBits[i]; mb->st->prevCode[i] = L.hufCode[i]; }
                if (zh_opaque(lane) == 0) { mb->st->prevRepeat = 1; mb->st->prevMaxSym = newMaxSym; }
            }
            // ZSTD_selectEncodingType's *repeatMode (zstd.c:21252): basic / rle -> none, compressed -> check, repeat keeps the dictionary's table valid
            if (fseRep && nbSeq && zh_opaque(lane) == 0)
                for (uint32_t t = 0; t < 3; t++) { const uint32_t md = (seqModes >> (2 * t)) & 3; if (md != 3) mb->st->fseRep[t] = md == 2 ? 1u : 0u; }
        }
        // "after the first block, the offcode table might not have large enough codes" (zstd.c:27392): valid -> check, whatever the block became
        if (fseRep && zh_opaque(lane) == 0 && mb->st->fseRep[1] == 2) mb->st->fseRep[1] = 1;
        ze_fence();
        zh_sync();
    }
    return r;
}

// ------------------------------------------------------------------------------------------ frame
// the row of the source's size class (level + explicit parameters, resolved on the host: ZSTD_getCParams_internal zstd.c:30848 +
// ZSTD_overrideCParams :24578) adjusted to a known source size without dictionary (ZSTD_adjustCParams_internal :24427)
ZH_DEV int ze_get_cparams(ZePar& out, const ZeRows& rows, uint32_t srcSize)
{
    const uint32_t tableID = (srcSize <= 256u * 1024) + (srcSize <= 128u * 1024) + (srcSize <= 16u * 1024);
    const int32_t* const r = rows.r[tableID];
    int w = r[0], c = r[1], h = r[2];
    const int srcLog = srcSize < 64 ? 6 : zh_highbit32(srcSize - 1) + 1;
    if (w > srcLog) w = srcLog;
    if (h > w + 1) h = w + 1;
    if (c > w) c = w;                                      // cycleLog == chainLog below btlazy2
    if (w < 10) w = 10;
    out.wlog = w; out.clog = c; out.hlog = h; out.mml = r[4]; out.strat = r[6]; out.tlen = r[5];
    if (out.strat != 1 && out.strat != 2) return ZE_PARAM_UNSUPPORTED;     // greedy and above are not implemented: refused, never approximated
    // an explicit window smaller than the source AND smaller than a block makes the match window slide inside a block
    // (ZSTD_getLowestPrefixIndex, zstd.c:19470); the level tables never produce that and the kernels do not implement it: refused
    if (w < 17 && (1u << w) < srcSize) return ZE_PARAM_UNSUPPORTED;
    // the search kernels' packed sequences (ZE_SEQ_PACK) carry offset + 3 in 28 bits: a window above 128 MiB (explicit window_log >= 28 on a
    // source that large) is refused rather than truncated
    if (w > 27) return ZE_PARAM_UNSUPPORTED;
    return 0;
}

// Where to end a full 128 KiB block once the frame has shown savings (ZSTD_splitBlock, zstd.c:22263-22502). All lanes call.
// fast strategy: byte histograms of the first / last / middle 512 bytes; double-fast: 8 KiB chunks, a byte histogram of every 43rd
// position, split at the first chunk whose histogram strays from everything before it.
ZH_DEV uint32_t ze_fp_distance(const uint32_t* A, uint32_t na, const uint32_t* B, uint32_t nb)     // sum |A[i] nb - B[i] na| over 256 bins (fits 32 bits)
{
    const uint32_t lane = zh_lane();
    uint32_t d = 0;
    for (uint32_t i = lane; i < 256; i += 64) { const int32_t x = (int32_t)(A[i] * nb) - (int32_t)(B[i] * na); d += (uint32_t)(x < 0 ? -x : x); }
    return zh_shfl(zh_scan_add(d), 63);
}
ZH_DEVFN uint32_t ze_split_block(ZeLDS& L, const uint8_t* p, int strat)
{
    const uint32_t lane = zh_lane();
    const uint32_t B = 128u << 10;
    uint32_t* const past = L.hist; uint32_t* const cur = (uint32_t*)L.node; uint32_t* const mid = cur + 256;
    zh_sync();
    for (uint32_t i = lane; i < 256; i += 64) { past[i] = 0; cur[i] = 0; mid[i] = 0; }
    zh_sync();
    if (strat == 1) {
        for (uint32_t i = lane; i < 512; i += 64) { zh_lds_atomic_inc(&past[p[i]]); zh_lds_atomic_inc(&cur[p[B - 512 + i]]); }
        zh_sync();
        const uint32_t dev = ze_fp_distance(past, 512, cur, 512);
        if (!(dev >= (uint32_t)(((uint64_t)512 * 512 * 14) / 16))) return B;
        for (uint32_t i = lane; i < 512; i += 64) zh_lds_atomic_inc(&mid[p[B / 2 - 256 + i]]);
        zh_sync();
        const uint32_t db = ze_fp_distance(past, 512, mid, 512), de = ze_fp_distance(cur, 512, mid, 512);
        const uint32_t diff = db > de ? db - de : de - db;
        if (diff < 512u * 512u / 3u) return 64u << 10;
        return db > de ? (32u << 10) : (96u << 10);
    }
    const uint32_t C = 8u << 10, nSamples = 191, nEv = 190;       // positions 0, 43, ... < 8191; the event count is 8191 / 43
    for (uint32_t k = lane; k < nSamples; k += 64) zh_lds_atomic_inc(&past[p[43 * k]]);
    zh_sync();
    uint32_t npast = nEv, penalty = 3;
    for (uint32_t pos = C; pos <= B - C; pos += C) {
        for (uint32_t i = lane; i < 256; i += 64) cur[i] = 0;
        zh_sync();
        for (uint32_t k = lane; k < nSamples; k += 64) zh_lds_atomic_inc(&cur[p[pos + 43 * k]]);
        zh_sync();
        const uint32_t dev = ze_fp_distance(past, npast, cur, nEv);
        const uint32_t threshold = (uint32_t)(((uint64_t)npast * nEv * (14 + penalty)) / 16);
        if (dev >= threshold) return pos;
        for (uint32_t i = lane; i < 256; i += 64) past[i] += cur[i];
        npast += nEv;
        if (penalty > 0) penalty--;
        zh_sync();
    }
    return B;
}

// The block layout of a source of several blocks for the flat match kernel (ZeMbBlock in zhip_format.hpp): ZSTD_compress_frameChunk's loop
// (zstd.c:27545) on the assumption that every block after the first is split-checked. One wave per frame; frames the flat search does not
// cover (dictionary, strategy, window below the source, tables above the slot, 4 MiB and more, more blocks than slots) get a count of 0.
ZH_DEVFN void ze_split_body(const ZhipEncodeArgs& a, ZeLDS& L)
{
    const uint32_t lane = zh_lane();
    for (uint32_t i = zh_block(); i < a.count; i += zh_nblocks()) {
        const uint32_t f = a.first + i;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
        uint32_t count = 0;
        ZePar cp;
        if (srcSize64 > ZF_BLOCK_MAX && srcSize64 < (1ull << ZE_MB_POS_BITS) - 8 && !a.cdict && ze_get_cparams(cp, a.rows, (uint32_t)srcSize64) == 0 &&
            cp.strat == 2 && (1ull << cp.wlog) >= srcSize64 && (size_t)(4u << cp.hlog) + (4u << cp.clog) <= a.tableStride &&
            // the sequence slice was sized from the caller's size HINT (srcSize / 4 + blocks + 64 sequences at most: matches are four bytes or
            // more); a source the hint understated would write past its slice -- it is the generic kernel's
            srcSize64 / 4 + a.mbMaxBlocks + 64 <= a.mbSeqCap) {
            const uint32_t srcSize = (uint32_t)srcSize64;
            ZeMbBlock* const blk = a.mbBlocks + (size_t)i * a.mbMaxBlocks;
            uint32_t ip = 0;
            while (ip < srcSize) {
                const uint32_t remaining = srcSize - ip;
                uint32_t blockSize = remaining < ZF_BLOCK_MAX ? remaining : ZF_BLOCK_MAX;
                if (remaining >= ZF_BLOCK_MAX && count >= 1) blockSize = ze_split_block(L, src + ip, cp.strat);
                ip += blockSize;
                if (count == a.mbMaxBlocks) { count = 0; break; }
                if (zh_opaque(lane) == 0) blk[count].end = ip;
                count++;
            }
        }
        zh_sync();
        if (zh_opaque(lane) == 0) a.mbCount[i] = count;
    }
    ze_fence();
}

// A frame of several blocks (ZSTD_compress_frameChunk, zstd.c:27545): sources above 128 KiB. All lanes call.
// useFlat: take the flat match kernel's sequences where it left some (ZeMbBlock, zhip_format.hpp) -- returns ZE_MB_RETRY when what really
// happened to the blocks is not what that search assumed; the caller then runs the frame again with useFlat = false.
#define ZE_MB_RETRY 0x7FFF0003
ZH_DEVFN int ze_frame_multi_impl(const ZhipEncodeArgs& a, ZeLDS& L, ZeLDSMulti* ms, uint32_t f, uint8_t* ws, uint64_t* produced, bool useFlat)
{
    const uint32_t lane = zh_lane();
    *produced = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    if (srcSize64 >= (1ull << 31)) return ZE_PARAM_UNSUPPORTED;                   // index overflow correction: not implemented
    const uint32_t srcSize = (uint32_t)srcSize64;
    if (cap64 < (uint64_t)srcSize + (srcSize >> 8)) return ZE_DST_TOO_SMALL;
    ZePar cp;
    const int e = ze_get_cparams(cp, a.rows, srcSize);
    if (e) return e;
    const ZeCDict* const cd = a.cdict;
    uint32_t dictID = 0;
    if (cd) {
        if (cd->status) return cd->status;
        if (cd->contentSize) {                                                     // table-copy mode; a window that would drop the dictionary part-way
            const int ce = ze_dict_copy_cparams(cp, *cd, a.rows, srcSize);        // (ZSTD_checkDictValidity, zstd.c:19360) is refused inside
            if (ce) return ce;
        }
        if (a.dictIDFlag) dictID = cd->dictID;
    }
    const uint32_t dictCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    if ((cp.strat != 2 && cp.strat != 1) || cp.hlog > ZE_MAX_HLOG || cp.clog > ZE_MAX_HLOG) return ZE_PARAM_UNSUPPORTED;
    const uint32_t contentSize = a.contentSizeFlag != 0, checksum = a.checksumFlag != 0;
    const uint32_t single = contentSize && cp.wlog < 32 && (1ull << cp.wlog) >= srcSize;
    const uint32_t fcsCode = contentSize ? (srcSize >= 256) + (srcSize >= 65536 + 256) : 0;
    uint32_t pos = 0;
    zh_sync();
    if (zh_opaque(lane) == 0) {
        if (!a.magicless) { zh_st32(dst, ZF_MAGIC); pos = 4; }
        dst[pos++] = (uint8_t)(dictCode + (checksum << 2) + (single << 5) + (fcsCode << 6));
        if (!single) dst[pos++] = (uint8_t)((cp.wlog - 10) << 3);
        if (dictCode == 1) dst[pos++] = (uint8_t)dictID;
        else if (dictCode == 2) { zh_st16(dst + pos, (uint16_t)dictID); pos += 2; }
        else if (dictCode == 3) { zh_st32(dst + pos, dictID); pos += 4; }
        if (fcsCode == 0) { if (single) dst[pos++] = (uint8_t)srcSize; }
        else if (fcsCode == 1) { zh_st16(dst + pos, (uint16_t)(srcSize - 256)); pos += 2; }
        else { zh_st32(dst + pos, srcSize); pos += 4; }
        L.misc[4] = pos;
        ms->mrep[0] = cd ? cd->rep[0] : 1; ms->mrep[1] = cd ? cd->rep[1] : 4; ms->prevMaxSym = cd ? cd->hufMaxSym : 0; ms->prevRepeat = cd ? cd->hufRepeat : 0;
        ms->fseRep[0] = cd ? cd->llRepeat : 0; ms->fseRep[1] = cd ? cd->ofRepeat : 0; ms->fseRep[2] = cd ? cd->mlRepeat : 0;
    }
    if (cd && cd->hufRepeat) for (uint32_t i = lane; i < 256; i += 64) { ms->prevBits[i] = cd->hufBits[i]; ms->prevCode[i] = cd->hufCode[i]; }
    zh_sync();
    pos = zh_first(L.misc[4]);
    zh_sync();
    ZeMulti mb; mb.frame = src; mb.firstBlock = true; mb.st = ms; mb.blkOff = 0;
    uint32_t ip = 0; int32_t savings = 0;
    // the flat match kernel's work on this frame, if any (chunk-local index: the kernel runs per chunk when that search is on)
    const uint32_t ci = f - a.first;
    const uint32_t nFlat = useFlat && a.mbCount && f >= a.first && ci < a.count && a.meta[ci].mode == 5 ? a.mbCount[ci] : 0u;
    const ZeMbBlock* const blk = nFlat ? a.mbBlocks + (size_t)ci * a.mbMaxBlocks : nullptr;
    uint32_t bj = 0;
    while (ip < srcSize) {
        const uint32_t remaining = srcSize - ip;
        uint32_t blockSize = remaining < ZF_BLOCK_MAX ? remaining : ZF_BLOCK_MAX;
        if (remaining >= ZF_BLOCK_MAX && savings >= 3) blockSize = nFlat && bj ? blk[bj].end - ip : ze_split_block(L, src + ip, cp.strat);   // (the split kernel's answer for this position)
        const uint32_t last = blockSize == remaining ? 1u : 0u;
        const uint64_t room = cap64 - pos - 3;                      // never let a block's scratch output run past this frame's slot
        const uint32_t want = blockSize + (blockSize >> 7) + 512;
        ZePre pre; const ZePre* prep = nullptr;
        if (nFlat) {
            // the search assumed: this block ends where the split kernel put it (it split-checked every block after the first), and starts
            // from the repeat offsets the block before ended with (every block confirmed, i.e. emitted compressed)
            const uint32_t s0 = bj ? blk[bj - 1].rep0 : 1u, s1 = bj ? blk[bj - 1].rep1 : 4u;
            if (bj >= nFlat || blk[bj].end != ip + blockSize || zh_first(ms->mrep[0]) != s0 || zh_first(ms->mrep[1]) != s1) return ZE_MB_RETRY;
            pre.seqs = a.mbSeqs + (size_t)ci * a.mbSeqCap + blk[bj].seqStart; pre.lits = nullptr; pre.nbSeq = blk[bj].nbSeq; pre.litSize = 0;
            prep = &pre;
            zh_sync();
            if (zh_opaque(lane) == 0) { L.misc[5] = blk[bj].rep0; L.misc[6] = blk[bj].rep1; }      // what ze_compress_block confirms when the block is emitted compressed
            zh_sync();
            bj++;
        }
        const uint32_t c = ze_compress_block<true>(L, dst + pos + 3, room < want ? (uint32_t)room : want, src + ip, blockSize, cp, ws, prep, a, &mb);
        uint32_t total, bh;
        if (c == 0) {
            bh = last + (0u << 1) + (blockSize << 3);
            ze_copy_wave(dst + pos + 3, src + ip, blockSize);
            total = 3 + blockSize;
        } else if (c == 1) { bh = last + (1u << 1) + (blockSize << 3); total = 4; }
        else { bh = last + (2u << 1) + (c << 3); total = 3 + c; }
        if (zh_opaque(lane) == 0) { dst[pos] = (uint8_t)bh; dst[pos + 1] = (uint8_t)(bh >> 8); dst[pos + 2] = (uint8_t)(bh >> 16); }
        savings += (int32_t)blockSize - (int32_t)total;
        pos += total; ip += blockSize; mb.firstBlock = false; mb.blkOff = ip;
        ze_fence();
        zh_sync();
    }
    if (checksum) {
        if (!a.xxLater && zh_opaque(lane) == 0) zh_st32(dst + pos, (uint32_t)ze_xxh64(src, srcSize));     // (xxLater: EX fills the trailer -- one lane hashing the source while 63 wait is a fifth of this kernel's time per frame)
        pos += 4;
    }
    ze_fence();
    *produced = pos;
    return ZE_OK;
}
ZH_DEVFN int ze_frame_multi(const ZhipEncodeArgs& a, ZeLDS& L, ZeLDSMulti* ms, uint32_t f, uint8_t* ws, uint64_t* produced)
{
    int r = ze_frame_multi_impl(a, L, ms, f, ws, produced, true);
    if (r == ZE_MB_RETRY) {
#ifdef ZHIP_EMU
        if (zh_lane() == 0) zd_stat[9]++;                                           // (test hook [9]: frames redone)
#endif
        zh_sync(); r = ze_frame_multi_impl(a, L, ms, f, ws, produced, false);
    }
    return r;
}

// one frame: header (ZSTD_writeFrameHeader zstd.c:27649), the block, optional checksum. All lanes call.
template <bool SEARCH>
ZH_DEVFN int ze_frame(const ZhipEncodeArgs& a, ZeLDS& L, uint32_t f, uint8_t* ws, uint64_t* produced, const ZePre* pre, ZeLDSMulti* ms = nullptr, ZeProf* P = nullptr)
{
    const uint32_t lane = zh_lane();
    *produced = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    if (srcSize64 > ZF_BLOCK_MAX) {                                    // multi-block frames: generic kernel only
        if constexpr (SEARCH) return (pre || !ms) ? ZE_PARAM_UNSUPPORTED : ze_frame_multi(a, L, ms, f, ws, produced);
        else return ZE_PARAM_UNSUPPORTED;
    }
    const uint32_t srcSize = (uint32_t)srcSize64;
    const uint32_t bound = srcSize + (srcSize >> 8) + (srcSize < (128u << 10) ? (((128u << 10) - srcSize) >> 11) : 0);
    if (cap64 < bound) return ZE_DST_TOO_SMALL;
    const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
    ZePar cp;
    const int e = ze_get_cparams(cp, a.rows, srcSize);
    if (e) return e;
    uint32_t dictID = 0;
    if (a.cdict) {
        if (a.cdict->status) return a.cdict->status;
        if (a.cdict->contentSize && srcSize > ze_dict_attach_max(*a.cdict)) {         // libzstd's table-copy mode: the generic kernel's work
            if constexpr (!SEARCH) return ZE_PARAM_UNSUPPORTED;
            const int ce = ze_dict_copy_cparams(cp, *a.cdict, a.rows, srcSize);
            if (ce) return ce;
        } else ze_dict_cparams(cp, *a.cdict, srcSize);
        if (a.dictIDFlag) dictID = a.cdict->dictID;
    }
    if ((cp.strat != 2 && cp.strat != 1) || cp.hlog > ZE_MAX_HLOG || cp.clog > ZE_MAX_HLOG) return ZE_PARAM_UNSUPPORTED;
    const uint32_t dictCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    uint32_t pos = 0;
    const uint32_t contentSize = a.contentSizeFlag != 0, checksum = a.checksumFlag != 0;
    const uint32_t windowSize = 1u << cp.wlog;
    const uint32_t single = contentSize && windowSize >= srcSize;
    const uint32_t fcsCode = contentSize ? (srcSize >= 256) + (srcSize >= 65536 + 256) : 0;
    zh_sync();
    if (zh_opaque(lane) == 0) {
        if (!a.magicless) { zh_st32(dst, ZF_MAGIC); pos = 4; }
        dst[pos++] = (uint8_t)(dictCode + (checksum << 2) + (single << 5) + (fcsCode << 6));
        if (!single) dst[pos++] = (uint8_t)((cp.wlog - 10) << 3);
        if (dictCode == 1) dst[pos++] = (uint8_t)dictID;
        else if (dictCode == 2) { zh_st16(dst + pos, (uint16_t)dictID); pos += 2; }
        else if (dictCode == 3) { zh_st32(dst + pos, dictID); pos += 4; }
        if (fcsCode == 0) { if (single) dst[pos++] = (uint8_t)srcSize; }
        else if (fcsCode == 1) { zh_st16(dst + pos, (uint16_t)(srcSize - 256)); pos += 2; }
        else { zh_st32(dst + pos, srcSize); pos += 4; }
        L.misc[4] = pos;
    }
    zh_sync();
    pos = zh_first(L.misc[4]);
    zh_sync();
    if (srcSize == 0) {
        if (zh_opaque(lane) == 0) { dst[pos] = 1; dst[pos + 1] = 0; dst[pos + 2] = 0; }
        pos += 3;
    } else {
        const uint32_t c = ze_compress_block<SEARCH>(L, dst + pos + 3, cap - pos - 3, src, srcSize, cp, ws, pre, a, nullptr, P);
        if (c == 0) {
            const uint32_t bh = 1 + (0u << 1) + (srcSize << 3);
            if (zh_opaque(lane) == 0) { dst[pos] = (uint8_t)bh; dst[pos + 1] = (uint8_t)(bh >> 8); dst[pos + 2] = (uint8_t)(bh >> 16); }
            ze_copy_wave(dst + pos + 3, src, srcSize);
            pos += 3 + srcSize;
        } else {
            const uint32_t bh = 1 + (2u << 1) + (c << 3);
            if (zh_opaque(lane) == 0) { dst[pos] = (uint8_t)bh; dst[pos + 1] = (uint8_t)(bh >> 8); dst[pos + 2] = (uint8_t)(bh >> 16); }
            pos += 3 + c;
        }
    }
    if (checksum) {
        if (!a.xxLater && zh_opaque(lane) == 0) zh_st32(dst + pos, (uint32_t)ze_xxh64(src, srcSize));     // (xxLater: EX fills the trailer -- one lane hashing the source while 63 wait is a fifth of this kernel's time per frame)
        pos += 4;
    }
    ze_fence();
    *produced = pos;
    return ZE_OK;
}


// ------------------------------------------------------------------------------------------ dictionary digestion (one wave, once per dictionary)
// parameters of the dictionary's own tables: ZSTD_getCParams_internal(level, unknown source, dictSize, ZSTD_cpm_createCDict)
// (zstd.c:30848, :24426)
ZH_DEV int ze_cdict_params(ZePar& out, const ZeRows& rows, uint32_t dictSize)
{
    const uint64_t rSize = (uint64_t)dictSize + 499;
    const uint32_t tableID = (rSize <= 256u * 1024) + (rSize <= 128u * 1024) + (rSize <= 16u * 1024);
    const int32_t* const r = rows.r[tableID];
    int w = r[0], c = r[1], h = r[2];
    const uint32_t srcSize = 513;                       // createCDict mode assumes a small source
    const uint32_t t = srcSize + dictSize;
    const int srcLog = t < 64 ? 6 : zh_highbit32(t - 1) + 1;
    if (w > srcLog) w = srcLog;
    int dawl = w;
    if ((1ull << w) < (uint64_t)dictSize + srcSize) { const uint64_t dw = (uint64_t)dictSize + (1ull << w); dawl = dw >= (1ull << 31) ? 31 : zh_highbit32((uint32_t)dw - 1) + 1; }
    if (h > dawl + 1) h = dawl + 1;
    if (c > dawl) c = dawl;
    if (w < 10) w = 10;
    if (h > 24) h = 24;
    if (c > 24) c = 24;
    out.wlog = w; out.clog = c; out.hlog = h; out.mml = r[4]; out.strat = r[6]; out.tlen = r[5];
    return 0;
}
ZH_DEV uint32_t ze_ncount_repeat(const int16_t* norm, uint32_t dictMax, uint32_t needMax)      // ZSTD_dictNCountRepeat, zstd.c:27998
{
    if (dictMax < needMax) return 1;
    for (uint32_t s = 0; s <= needMax; s++) if (norm[s] == 0) return 1;
    return 2;
}

// ZSTD_loadCEntropy (zstd.c:28015) from the already-parsed entropy section (de, null for a raw-content dictionary), then
// ZSTD_fillDoubleHashTableForCDict (:30952) done wave-parallel: the sequential fill keeps, per cell, the LAST position of the
// every-third-position series and otherwise the FIRST of the in-between positions, which is an atomic max / min.
// hashLong, hashSmall: 1 << ZE_CDICT_MAX_HLOG cells each; tmpLong: scratch of the same size. All lanes call.
ZH_DEVFN void ze_cdict_body(const uint8_t* dict, uint32_t dictSize, const ZhipDictEntropy* de, const ZeRows& rows, ZeCDict* cd,
                            uint32_t* hashLong, uint32_t* hashSmall, uint32_t* tmpLong, ZeLDS& L)
{
    const uint32_t lane = zh_lane();
    const bool hasEntropy = de && de->hufCount != 0;
    const uint32_t contentOff = hasEntropy ? de->contentOffset : 0u;
    const uint32_t cs = dictSize - contentOff;
    const uint8_t* content = dict + contentOff;
    zh_sync();
    if (zh_opaque(lane) == 0) {
        ZePar p; p.wlog = p.clog = p.hlog = p.mml = p.strat = p.tlen = 0;
        int st = ze_cdict_params(p, rows, dictSize);
        if (!st && ((p.strat != 2 && p.strat != 1) || p.hlog > ZE_CDICT_MAX_HLOG || (p.strat == 2 && p.clog > ZE_CDICT_MAX_HLOG) || cs > ZE_CDICT_MAX_CONTENT)) st = ZE_PARAM_UNSUPPORTED;
        cd->hlog = p.hlog; cd->clog = p.clog; cd->mml = p.mml; cd->strat = p.strat; cd->tlen = p.tlen;
        cd->contentSize = dictSize < 8 ? 0u : cs; cd->dictID = hasEntropy ? de->dictID : 0u;
        cd->rep[0] = 1; cd->rep[1] = 4; cd->rep[2] = 8;
        cd->hufRepeat = cd->llRepeat = cd->ofRepeat = cd->mlRepeat = 0; cd->hufMaxSym = 0;
        L.misc[5] = 0;
        if (!st && hasEntropy) {
            // HUF_readCTable (zstd.c:17048): code lengths from weights, canonical values per length in symbol order
            const uint32_t cnt = de->hufCount;
            uint32_t total = 0; bool zero = false;
            for (uint32_t s = 0; s < cnt; s++) { const uint32_t w = de->hufWeights[s]; total += w ? (1u << w) >> 1 : 0u; zero |= (w == 0); }
            const uint32_t lg = total ? (uint32_t)zh_highbit32(total) : 0u;
            if (lg == 0 || lg > 12) st = ZE_DICT_CORRUPTED;
            else {
                uint16_t perRank[16], start[16];
                for (int r = 0; r < 16; r++) { perRank[r] = 0; start[r] = 0; }
                for (uint32_t s = 0; s < 256; s++) { cd->hufBits[s] = 0; cd->hufCode[s] = 0; }
                for (uint32_t s = 0; s < cnt; s++) { const uint32_t w = de->hufWeights[s]; const uint32_t nb = w ? lg + 1 - w : 0u; cd->hufBits[s] = (uint8_t)nb; perRank[nb]++; }
                { uint16_t mn = 0; for (int r = (int)lg; r > 0; r--) { start[r] = mn; mn = (uint16_t)((mn + perRank[r]) >> 1); } }
                for (uint32_t s = 0; s < cnt; s++) { const uint32_t nb = cd->hufBits[s]; cd->hufCode[s] = nb ? start[nb]++ : (uint16_t)0; }
                cd->hufMaxSym = cnt - 1;
                cd->hufRepeat = (!zero && cnt == 256) ? 2u : 1u;
                {   const uint32_t need = (uint32_t)zh_highbit32(cs + 128u * 1024);
                    cd->ofRepeat = ze_ncount_repeat(de->ofNorm, de->ofMax, need < ZF_MAXOFF ? need : ZF_MAXOFF); }
                cd->mlRepeat = ze_ncount_repeat(de->mlNorm, de->mlMax, ZF_MAXML);
                cd->llRepeat = ze_ncount_repeat(de->llNorm, de->llMax, ZF_MAXLL);
                for (int i = 0; i < 3; i++) cd->rep[i] = de->rep[i];
                L.misc[5] = 1;                                                      // the three encoding tables follow, built by the whole wave
            }
        }
        cd->status = st;
        L.misc[0] = (uint32_t)st; L.misc[1] = (uint32_t)p.hlog; L.misc[2] = (uint32_t)p.clog; L.misc[3] = (uint32_t)p.mml; L.misc[4] = (uint32_t)p.strat;
    }
    ze_fence();
    zh_sync();
    const uint32_t st = zh_first(L.misc[0]);
    const int hlog = (int)zh_first(L.misc[1]), clog = (int)zh_first(L.misc[2]), mml = (int)zh_first(L.misc[3]);
    const bool fast = zh_first(L.misc[4]) == 1;          // ZSTD_fillHashTableForCDict (zstd.c:31730): ONE table, hashed on minMatch bytes, same fill rule as the long table
    const bool tables = zh_first(L.misc[5]) != 0;
    zh_sync();
    if (st) return;
    if (tables) {                                        // FSE encoding tables of the dictionary's distributions (all offset codes, like the reference)
        int16_t* const norm = ze_norm_area(L);
        for (int w = 0; w < 3; w++) {
            const int16_t* src = w == 0 ? de->llNorm : w == 1 ? de->ofNorm : de->mlNorm;
            const uint32_t mx = w == 0 ? de->llMax : w == 1 ? de->ofMax : de->mlMax, lim = w == 0 ? 36u : w == 1 ? 32u : 53u;
            norm[lane] = lane <= mx && lane < lim ? src[lane] : (int16_t)0;
            ze_fence(); zh_sync();
            ze_fse_build_ctab_wave(L.tab[w], norm, w == 1 ? (uint32_t)ZF_MAXOFF : mx, w == 0 ? de->llLog : w == 1 ? de->ofLog : de->mlLog, ze_cell_sym(L), (uint16_t*)L.stack, ze_fill_area(L));
            ze_fence(); zh_sync();
            {   const uint32_t* s4 = (const uint32_t*)&L.tab[w]; uint32_t* d4 = (uint32_t*)&cd->tab[w];       // built in LDS, copied out as dwords
                for (uint32_t i = lane; i < sizeof(ZeCTab) / 4; i += 64) d4[i] = s4[i]; }
            ze_fence(); zh_sync();
        }
    }
    for (uint32_t i = lane; i < (1u << hlog); i += 64) { hashLong[i] = 0; tmpLong[i] = 0; }
    if (!fast) for (uint32_t i = lane; i < (1u << clog); i += 64) hashSmall[i] = 0;
    ze_fence();
    zh_sync();
    // only the tail the tables can reasonably address is indexed (ZSTD_loadDictionaryContent, zstd.c:27895)
    uint32_t startOff = 0;
    {   int mx = hlog + 3 > clog + 1 ? hlog + 3 : clog + 1; if (mx > 31) mx = 31;
        const uint32_t maxDict = 1u << mx;
        if (cs > maxDict) startOff = cs - maxDict; }
    if (cs - startOff >= 10) {
        const int mls = mml <= 4 ? 4 : mml >= 7 ? 7 : mml;
        const uint32_t nGroups = (cs - 10 - startOff) / 3 + 1;        // group g covers positions startOff + 3g + {0,1,2}
        for (uint32_t g = lane; g < nGroups; g += 64) {
            const uint32_t pos = startOff + 3 * g, curr = pos + 2;
            const uint32_t lg = ze_hash(content + pos, hlog + 8, fast ? mls : 8);
            if (!fast) { const uint32_t sm = ze_hash(content + pos, clog + 8, mls); zh_atomic_max(&hashSmall[sm >> 8], (curr << 8) | (sm & 255)); }
            zh_atomic_max(&hashLong[lg >> 8], (curr << 8) | (lg & 255));
        }
        ze_fence();
        zh_sync();
        for (uint32_t g = lane; g < nGroups; g += 64) {
            for (uint32_t i = 1; i < 3; i++) {
                const uint32_t pos = startOff + 3 * g + i, curr = pos + 2;
                const uint32_t lg = ze_hash(content + pos, hlog + 8, fast ? mls : 8);
                if (hashLong[lg >> 8] == 0) zh_atomic_max(&tmpLong[lg >> 8], ((0xFFFFFFu - curr) << 8) | (lg & 255));
            }
        }
        ze_fence();
        zh_sync();
        for (uint32_t i = lane; i < (1u << hlog); i += 64) {
            const uint32_t t = tmpLong[i];
            if (hashLong[i] == 0 && t) hashLong[i] = ((0xFFFFFFu - (t >> 8)) << 8) | (t & 255);
        }
        ze_fence();
        zh_sync();
    }
}

ZH_DEVFN void ze_kernel_body(const ZhipEncodeArgs& a, ZeLDS& L, ZeLDSMulti& M)
{
    const uint32_t lane = zh_lane();
    uint8_t* ws = a.workspace + (size_t)zh_block() * ZHIP_ENC_STRIDE;
    const uint32_t total = a.frameList ? *a.listCount : a.n;              // an explicit list: the inputs the two-kernel form declined
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counter, lane == 0 ? 1u : 0u);     // branch-free fetch (see decoder note)
        if (zh_opaque(lane) == 0) L.misc[15] = got;
        zh_sync();
        const uint32_t k = zh_first(L.misc[15]);
        zh_sync();
        if (k >= total) break;
        const uint32_t f = a.frameList ? a.frameList[k] : k;
        uint64_t produced = 0;
        const int err = ze_frame<true>(a, L, f, ws, &produced, nullptr, &M);
        zh_sync();
        if (zh_opaque(lane) == 0) { a.status[f] = err; a.outSizes[f] = err ? 0 : produced; }
    }
}

// ------------------------------------------------------------------------------------------ two-kernel form
// E1: double-fast search with one LANE per frame. The search is a chain of dependent global-memory probes (hash table,
// candidate bytes), so the way to throughput is frames in flight: every lane of the wave runs the serial search of its own
// frame against its own tables; nothing is shared between lanes and no cross-lane operation is needed.
ZH_DEVFN void ze_match_body(const ZhipEncodeArgs& a)
{
    const uint32_t lane = zh_lane();
    if (lane >= a.e1Lanes) return;
    uint8_t* tables = a.laneTables + ((size_t)zh_block() * a.e1Lanes + lane) * a.tableStride;
    for (;;) {
        const uint32_t k = zh_atomic_add(a.counter, 1u);
        if (k >= (a.useE1List ? *a.e1Count : a.count)) break;
        const uint32_t i = a.useE1List ? a.e1List[k] : k;
        const uint32_t f = a.first + i;
        ZeMeta m; m.nbSeq = 0; m.litSize = 0; m.mode = 0; m.pad = 0;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
        ZePar cp;
        if (srcSize64 > ZF_BLOCK_MAX) {                                    // several blocks: the generic kernel takes it
            m.mode = 3; a.meta[i] = m;
            a.bigList[zh_atomic_add(a.bigCount, 1u)] = f;
            continue;
        }
        const uint32_t srcSize = (uint32_t)srcSize64;
        bool bad = ze_get_cparams(cp, a.rows, srcSize) != 0;
        if (!bad && a.cdict) {
            if (a.cdict->status) bad = true;
            else if (a.cdict->contentSize && srcSize > ze_dict_slot_max(a)) {            // table-copy mode (needs tables of the dictionary's own size and a
                m.mode = 3; a.meta[i] = m;                                              // 128 KiB arena slot): one wave per frame in the generic kernel
                a.bigList[zh_atomic_add(a.bigCount, 1u)] = f;
                continue;
            }
            else ze_dict_cparams(cp, *a.cdict, srcSize);
        }
        if (bad || (cp.strat != 2 && cp.strat != 1) ||
            (size_t)(4u << cp.hlog) + (cp.strat == 2 ? (4u << cp.clog) : 0u) > a.tableStride) { m.mode = 2; a.meta[i] = m; continue; }   // E2 reports the error
        if (srcSize < 7) { m.mode = 1; a.meta[i] = m; continue; }
        uint32_t* hashLong = (uint32_t*)tables;
        uint32_t* hashSmall = (uint32_t*)(tables + (4u << cp.hlog));
        { ZdPack16* z = (ZdPack16*)tables; const uint32_t nz = ((4u << cp.hlog) + (cp.strat == 2 ? (4u << cp.clog) : 0u)) / 16; ZdPack16 zero; zero.a = zero.b = zero.c = zero.d = 0; for (uint32_t k = 0; k < nz; k++) z[k] = zero; }
        uint8_t* fr = a.arena + (size_t)i * a.arenaStride;
        uint32_t litSize = 0;
        // a dictionary without content (shorter than 8 bytes: nothing of it is loaded, zstd.c:28167) is not attached
        // (ZSTD_resetCCtx_byAttachingCDict, "don't even attach dictionaries with no contents"): the plain search, with the dictionary's row
        m.nbSeq = (a.cdict && a.cdict->contentSize) ? (cp.strat == 1 ? ze_fast_dict((uint64_t*)(fr + ZE_ARENA_SEQ), fr + a.arenaLit, &litSize, src, srcSize, cp, *a.cdict, a.cdictContent,
                                                                                   a.cdictHashLong, hashLong)
                                                                      : ze_dfast_dict((uint64_t*)(fr + ZE_ARENA_SEQ), fr + a.arenaLit, &litSize, src, srcSize, cp, *a.cdict, a.cdictContent,
                                                                                      a.cdictHashLong, a.cdictHashSmall, hashLong, hashSmall))
                          : cp.strat == 1 ? ze_fast((uint64_t*)(fr + ZE_ARENA_SEQ), nullptr, &litSize, src, srcSize, cp, hashLong)
                          : ze_dfast((uint64_t*)(fr + ZE_ARENA_SEQ), fr + a.arenaLit, &litSize, src, srcSize, cp, hashLong, hashSmall);
        m.litSize = litSize;
        if (!(a.cdict && a.cdict->contentSize) && cp.strat == 1) m.mode = 4;                  // sequences only: the entropy kernel gathers the literals
        a.meta[i] = m;
    }
}

// E1 flat: one lane per frame, statically assigned, every frame of the chunk in flight (ze_dfast_flat). Frames it does not cover are
// listed for the lane-serial kernel above (chunk-local index) or, above one block, for the generic kernel.
template <int NPROBE = 2>
ZH_DEVFN void ze_match_flat_body(const ZhipEncodeArgs& a)
{
    const uint32_t lane = zh_lane();
    const uint32_t i = zh_block() * ZE_FLAT_LANES + lane;
    const bool mine = lane < ZE_FLAT_LANES && i < a.count;
    const uint32_t f = a.first + (mine ? i : 0u);
    ZeMeta m; m.nbSeq = 0; m.litSize = 0; m.mode = 0; m.pad = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = mine ? a.srcSegs[2 * (size_t)f + 1] : 0;
    const uint32_t srcSize = (uint32_t)srcSize64;
    ZePar cp; cp.wlog = cp.clog = cp.hlog = cp.mml = cp.strat = cp.tlen = 0;
    bool take = mine;
    if (mine && srcSize64 > ZF_BLOCK_MAX) {                       // (sources the split kernel laid out are ze_match_flat_mb_body's: it lists them itself)
        if (!(a.mbCount && a.mbCount[i])) { m.mode = 3; a.meta[i] = m; a.bigList[zh_atomic_add(a.bigCount, 1u)] = f; }
        take = false;
    }
    // with an attached dictionary (sources up to the attach cutoff, double-fast row, a dictionary that has content): ze_dfast_dict_flat
    const bool dict = a.cdict != nullptr;
    if (take) {
        bool ok = srcSize >= 64 && ze_get_cparams(cp, a.rows, srcSize) == 0;
        if (ok && dict) {
            ok = a.cdict->status == 0 && a.cdict->contentSize != 0 && srcSize <= ze_dict_slot_max(a);
            if (ok) ze_dict_cparams(cp, *a.cdict, srcSize);
        }
        ok = ok && cp.strat == 2 && (size_t)(4u << cp.hlog) + (4u << cp.clog) <= a.tableStride;
        if (!ok) { a.e1List[zh_atomic_add(a.e1Count, 1u)] = i; take = false; }    // the lane-serial kernel decides (and reports errors)
    }
    uint32_t* hashLong = (uint32_t*)(a.flatTables + (size_t)(mine ? i : 0u) * a.tableStride);
    uint32_t* hashSmall = hashLong + (take ? (1u << cp.hlog) : 0u);
    if (dict && a.tabEpoch == 0) {
        // dictionary batches without launch numbers in the cells (ZhipEncodeArgs.tabEpoch): the wave zeroes its documents' tables itself, only the part each one uses (the
        // slots are sized for the attach cutoff: a host-side memset of whole slots would write four times what 4 KiB documents need)
        const uint32_t myUnits = take ? ((4u << cp.hlog) + (4u << cp.clog)) / 16u : 0u;
        for (uint32_t l = 0; l < ZE_FLAT_LANES; l++) {
            const uint32_t units = zh_shfl(myUnits, l);
            if (!units) continue;
            const uint32_t lo = zh_shfl((uint32_t)(uintptr_t)hashLong, l), hi = zh_shfl((uint32_t)((uintptr_t)hashLong >> 32), l);
            ZdPack16* t = (ZdPack16*)(uintptr_t)(((uint64_t)hi << 32) | lo);
            ZdPack16 z; z.a = z.b = z.c = z.d = 0;
            for (uint32_t k = lane; k < units; k += 64) t[k] = z;
        }
        zd_fence();
        zh_sync();
    }
    if (!take) return;
    uint8_t* fr = a.arena + (size_t)i * a.arenaStride;
    // (a probe launch of the placement pick searches the sources' first bytes only -- same tables, same parameters, the same scatter over the allocation, a fraction of the time)
    const uint32_t searchSize = a.probeCap && srcSize > a.probeCap ? a.probeCap : srcSize;
    const uint64_t wc0 = a.waveClock ? zd_wall_clock() : 0ull;
    m.nbSeq = dict ? ze_dfast_dict_flat((uint64_t*)(fr + ZE_ARENA_SEQ), src, searchSize, cp, *a.cdict, a.cdictContent, a.cdictHashLong, a.cdictHashSmall, hashLong, hashSmall, a.tabEpoch, a.tabEpochShift)
                   : NPROBE > 2 ? ze_dfast_flatn<NPROBE>((uint64_t*)(fr + ZE_ARENA_SEQ), src, searchSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, a.idle, a.tabEpoch)
                                 : ze_dfast_flat((uint64_t*)(fr + ZE_ARENA_SEQ), src, searchSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, a.idle, a.tabEpoch);
    if (a.waveClock && lane == 0) a.waveClock[zh_block()] = zd_wall_clock() - wc0;
    m.mode = 4;
#ifdef ZHIP_EMU
    zd_stat[15]++;
#endif
    a.meta[i] = m;
}


// The flat search on sources of SEVERAL BLOCKS (ZeMbBlock in zhip_format.hpp): one lane per source the split kernel laid out, over all its
// blocks -- the hash tables (the frame's slot of the flat tables, zeroed by the host) and the repeat offsets carry from block to block as in
// ZSTD_compress_frameChunk (zstd.c:27545). A kernel of its own: the single-block kernel's registers and code stay what they were.
ZH_DEVFN void ze_match_flat_mb_body(const ZhipEncodeArgs& a)
{
    const uint32_t lane = zh_lane();
    const uint32_t i = zh_block() * a.mbLanes + lane;
    if (!(lane < a.mbLanes && i < a.count)) return;
    const uint32_t nb = a.mbCount[i];
    if (!nb) return;
    const uint32_t f = a.first + i;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint32_t srcSize = (uint32_t)a.srcSegs[2 * (size_t)f + 1];
    ZePar cp;
    ze_get_cparams(cp, a.rows, srcSize);                            // (the split kernel checked it)
    uint32_t* const hl = (uint32_t*)(a.flatTables + (size_t)i * a.tableStride);
    ZeMbBlock* const blk = a.mbBlocks + (size_t)i * a.mbMaxBlocks;
    uint64_t* const sq = a.mbSeqs + (size_t)i * a.mbSeqCap;
    uint32_t rep[2] = {1, 4}, bs = 0, start = 0;
    for (uint32_t j = 0; j < nb; j++) {
        const uint32_t be = blk[j].end;
        // (blocks too small to compress are not searched at all: ZSTD_buildSeqStore, zstd.c:26335)
        const uint32_t ns = be - bs < 7 ? 0u
                          : a.mbProbes == 4 ? ze_dfast_flat_np<ZE_MB_POS_BITS, true, 4>(sq + start, src, bs, be, cp.hlog, cp.clog, cp.mml, hl, hl + (1u << cp.hlog), rep, a.idle ? a.idle : src)
                                            : ze_dfast_flat_t<ZE_MB_POS_BITS, true>(sq + start, src, bs, be, cp.hlog, cp.clog, cp.mml, hl, hl + (1u << cp.hlog), rep);
        blk[j].seqStart = start; blk[j].nbSeq = ns; blk[j].rep0 = rep[0]; blk[j].rep1 = rep[1];
        start += ns; bs = be;
    }
    ZeMeta m; m.nbSeq = 0; m.litSize = 0; m.mode = 5; m.pad = 0;
#ifdef ZHIP_EMU
    zd_stat[8]++;                                                   // (test hook [8]: sources of several blocks searched here)
#endif
    a.meta[i] = m; a.bigList[zh_atomic_add(a.bigCount, 1u)] = f;
}

// E1 for SMALL batches (one-shot compress(), a few hundred frames): the flat kernel's search with the frame's source in LDS. With few
// frames nothing hides a probe's latency, and the search is a chain of dependent round trips -- its own bytes, the table cells, the
// candidates' bytes, then match extension / catch-up / the insertions' bytes, each a round trip of its own. One wave per frame copies
// the source (<= one block) into LDS, after which only the table cells are global: every other round becomes an LDS read. Same search
// function, same cells, same sequences -- only where the source bytes are read from differs.
// The LDS area is sized for the batch's largest source where the caller knows it (the host-buffer API does): small sources leave room
// for more frames per CU (4 KiB: 32 waves, 16 KiB: 9, 64 KiB: 2, one block: 1). A source above the area (no size hint and a shape picked
// too small cannot happen -- the host falls back to the one-block shape -- but the kernel does not rely on it) is searched in place.
template <uint32_t BYTES> struct ZeSrcLDS { uint8_t b[BYTES + 64]; };
template <int NPROBE = 2>
ZH_DEVFN void ze_match_lds_body(const ZhipEncodeArgs& a, uint8_t* lds, uint32_t ldsBytes)
{
    const uint32_t lane = zh_lane();
    const uint32_t i = zh_block();
    if (i >= a.count) return;
    const uint32_t f = a.first + i;
    ZeMeta m; m.nbSeq = 0; m.litSize = 0; m.mode = 0; m.pad = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    ZePar cp;
    if (srcSize64 > ZF_BLOCK_MAX) { if (lane == 0) { m.mode = 3; a.meta[i] = m; a.bigList[zh_atomic_add(a.bigCount, 1u)] = f; } return; }
    const uint32_t srcSize = (uint32_t)srcSize64;
    if (a.cdict || srcSize < 64 || ze_get_cparams(cp, a.rows, srcSize) || cp.strat != 2 ||
        (size_t)(4u << cp.hlog) + (4u << cp.clog) > a.tableStride) {
        if (lane == 0) a.e1List[zh_atomic_add(a.e1Count, 1u)] = i;        // the lane-serial kernel decides (and reports errors)
        return;
    }
    const bool staged = srcSize <= ldsBytes;
    if (staged) {
        const uint32_t whole = srcSize & ~15u;
        for (uint32_t k = lane * 16u; k < whole; k += 64u * 16u) {
            const zh_v16 v = zh_ld128(src + k);
            *(uint64_t*)(lds + k) = v.lo; *(uint64_t*)(lds + k + 8) = v.hi;
        }
        if (whole + lane < srcSize) lds[whole + lane] = src[whole + lane];     // the last partial unit byte-wise: nothing is read past the source
    }
    zh_sync();
    uint32_t* hashLong = (uint32_t*)(a.flatTables + (size_t)i * a.tableStride);
    uint32_t* hashSmall = hashLong + (1u << cp.hlog);
    uint8_t* fr = a.arena + (size_t)i * a.arenaStride;
    // (one lane searches. Round 3 also tried the whole wave on one source -- lanes 0..14 take the next 15 probes of a no-match stretch, one table
    // round trip for all of them, forwarding of the earlier lanes' inserts, the lowest lane with a hit is the reference's next match; bit-exact,
    // and SLOWER: one-shot 128 KiB 40.5 ms against 16.0, 256 inputs 60.7 against 47.1 (r03zb; commit f4216c1 has the code). A match ends the
    // stretch every 4.5 probes on the bench corpus and the one-lane search already takes two probes per round trip, so the wave form saves
    // little more than half the round trips and pays shuffles, ballots and a wave-wide match extension for each.)
    if (lane != 0) return;
    if (NPROBE == 4)
        m.nbSeq = staged ? ze_dfast_flat4((uint64_t*)(fr + ZE_ARENA_SEQ), lds, srcSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, nullptr, a.tabEpoch)
                         : ze_dfast_flat4((uint64_t*)(fr + ZE_ARENA_SEQ), src, srcSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, a.idle, a.tabEpoch);
    else
    m.nbSeq = staged ? ze_dfast_flat((uint64_t*)(fr + ZE_ARENA_SEQ), lds, srcSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, nullptr, a.tabEpoch)
                     : ze_dfast_flat((uint64_t*)(fr + ZE_ARENA_SEQ), src, srcSize, cp.hlog, cp.clog, cp.mml, hashLong, hashSmall, a.idle, a.tabEpoch);
    m.mode = 4;
#ifdef ZHIP_EMU
    zd_stat[15]++;
#endif
    a.meta[i] = m;
}

// E2: everything after the search (entropy coding + frame assembly), one wave per frame
ZH_DEVFN void ze_entropy_body(const ZhipEncodeArgs& a, ZeLDS& L)
{
    const uint32_t lane = zh_lane();
    // only the literal region of the fused kernel's workspace layout is used here; bias the base so it lands in our slot
    uint8_t* ws = a.workspace + (size_t)zh_block() * ZE_E2_STRIDE - ZE_WS_LIT;
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counter + 1, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[15] = got;
        zh_sync();
        const uint32_t i = zh_first(L.misc[15]);
        zh_sync();
        if (i >= a.count) break;
        const uint32_t f = a.first + i;
        const ZeMeta m = a.meta[i];
        if (m.mode == 3 || m.mode == 5) continue;                          // listed for the generic kernel
        const uint8_t* fr = a.arena + (size_t)i * a.arenaStride;
        ZePre pre; pre.seqs = (const uint64_t*)(fr + ZE_ARENA_SEQ); pre.lits = m.mode == 4 ? nullptr : fr + a.arenaLit; pre.nbSeq = m.nbSeq; pre.litSize = m.litSize;
        uint64_t produced = 0;
        ZeProf prof; ZeProf* P = a.prof ? &prof : nullptr;
        if (P) { for (int q = 0; q < ZEP_N; q++) prof.acc[q] = 0; prof.t0 = zd_clock(); }
        const int err = ze_frame<false>(a, L, f, ws, &produced, (m.mode == 0 || m.mode == 4) ? &pre : nullptr, nullptr, P);   // modes 1/2 never reach the search (raw / error)
        if (P) { ZE_T(P, ZEP_REST); if (zh_opaque(lane) == 0) for (int q = 0; q < ZEP_N; q++) zh_atomic_add64(a.prof + q, (unsigned long long)prof.acc[q]); }
        zh_sync();
        if (zh_opaque(lane) == 0) { a.status[f] = err; a.outSizes[f] = err ? 0 : produced; }
    }
}

// ------------------------------------------------------------------------------------------ EX (a LANE per frame: the frames' checksum trailers)
// write_checksum frames end with the low 32 bits of XXH64 of the SOURCE (zstd.c:28325-28329): a serial chain of 4 096 steps per 128 KiB that the entropy kernel ran on one lane
// of the frame's wave. With ZhipEncodeArgs.xxLater the entropy kernel only reserves the 4 bytes; here every lane hashes a source of its own and stores the trailer at the end
// of the frame that kernel produced. Frames listed for the generic kernel (modes 3 / 5) keep its own hashing.
ZH_DEVFN void ze_trailer_body(const ZhipEncodeArgs& a)
{
    const uint32_t lane = zh_lane();
    for (uint32_t i = zh_block() * 64 + lane; i < a.count; i += zh_nblocks() * 64) {
        const uint32_t mode = a.meta[i].mode;
        if (mode == 3 || mode == 5) continue;
        const uint32_t f = a.first + i;
        if (a.status[f] != 0) continue;
        const uint64_t n = a.outSizes[f];
        if (n < 4) continue;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize = a.srcSegs[2 * (size_t)f + 1];
        zh_st32(a.dst + a.dstSegs[2 * (size_t)f] + n - 4, (uint32_t)ze_xxh64(src, (uint32_t)srcSize));
    }
}
