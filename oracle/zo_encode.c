/* oracle/zo_encode.c -- CPU oracle: zstd frame ENCODER, bit-exact with libzstd 1.5.7 for the double-fast strategy
 * (TEST INFRASTRUCTURE ONLY, see zo_common.h).
 *
 * Restates, per SURVEY.md 8(a) / Appendix A, the compress side of the reference hot path for ONE frame of at most
 * one block (srcSize <= 128 KiB), which is what every batch configuration in BASELINE.json feeds the GPU:
 *   parameters        <- zstd.c:30848 ZSTD_getCParams_internal, :24426 ZSTD_adjustCParams_internal, tables :30650-30755
 *   frame header      <- zstd.c:27649 ZSTD_writeFrameHeader, epilogue :28298
 *   match finder      <- zstd.c:31039 ZSTD_compressBlock_doubleFast_noDict_generic (+ :20052-20099 hashes, :20008 ZSTD_count)
 *   sequence codes    <- zstd.c:25647 ZSTD_seqToCodes, :19738 ZSTD_LLcode, :19755 ZSTD_MLcode
 *   table modes       <- zstd.c:21252 ZSTD_selectEncodingType, :21338 ZSTD_buildCTable, :25716 ZSTD_buildSequencesStatistics
 *   FSE               <- zstd.c:16294 FSE_optimalTableLog, :16402 FSE_normalizeCount (+ :16316 M2), :16170 FSE_writeNCount,
 *                        :16005 FSE_buildCTable_wksp, :2774 FSE_initCState2, :2785 FSE_encodeSymbol
 *   sequence stream   <- zstd.c:21386 ZSTD_encodeSequences_body
 *   literals          <- zstd.c:20932 ZSTD_compressLiterals, :18089 HUF_compress_internal, :17377 HUF_sort, :17438 HUF_buildTree,
 *                        :17133 HUF_setMaxHeight, :17487 HUF_buildCTableFromTree, :17005 HUF_writeCTable_wksp,
 *                        :16904 HUF_compressWeights, :16488 FSE_compress_usingCTable_generic, :17925 HUF 4-stream layout
 *   block assembly    <- zstd.c:25842 ZSTD_entropyCompressSeqStore_internal, :25960 (minGain), :27337 ZSTD_compressBlock_internal
 * Every tie-break and threshold that decides output bytes is kept; data structures and code organisation are ours
 * (positions instead of pointers, explicit bit writer, sorted cell lists instead of libzstd's packed CTable layout).
 * Out of scope here (returns an error): multi-block frames (> 128 KiB), dictionaries, strategies other than dfast.
 */
#include "zstd_oracle.h"
#include "zo_common.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ parameters */
typedef struct { int wlog, clog, hlog, slog, mml, tlen, strat; } zo_cpar;   /* strat: 1 fast, 2 dfast, 3+ others */

/* level rows 1..4 of the four size classes (zstd.c:30650-30755); level 3 is the path of record */
static const zo_cpar zo_rows[4][5] = {
    /* > 256 KB */ {{19,12,13,1,6,1,1},{19,13,14,1,7,0,1},{20,15,16,1,6,0,1},{21,16,17,1,5,0,2},{21,18,18,1,5,0,2}},
    /* <= 256 KB*/ {{18,12,13,1,5,1,1},{18,13,14,1,6,0,1},{18,14,14,1,5,0,2},{18,16,16,1,4,0,2},{18,16,17,3,5,2,3}},
    /* <= 128 KB*/ {{17,12,12,1,5,1,1},{17,12,13,1,6,0,1},{17,13,15,1,5,0,1},{17,15,16,2,5,0,2},{17,17,17,2,4,0,2}},
    /* <= 16 KB */ {{14,12,13,1,5,1,1},{14,14,15,1,5,0,1},{14,14,15,1,4,0,1},{14,14,15,2,4,0,2},{14,14,14,4,4,2,3}},
};

static int zo_get_cparams(zo_cpar* out, int level, uint64_t srcSize)
{
    if (level == 0) level = 3;
    if (level > 4) return -ZO_E_PARAM_UNSUPPORTED;
    unsigned tableID = (srcSize <= 256u * 1024) + (srcSize <= 128u * 1024) + (srcSize <= 16u * 1024);
    zo_cpar c = zo_rows[tableID][level < 0 ? 0 : level];
    if (level < 0) {                                    /* negative levels: row 0 with targetLength = -level (zstd.c:30870-30875) */
        if (level < -(1 << 17)) level = -(1 << 17);
        c.tlen = -level;
    }
    /* shrink to the source (no dictionary): window, then hash and chain logs follow the window */
    uint32_t t = (uint32_t)srcSize;
    int srcLog = (t < 64) ? 6 : zo_highbit(t - 1) + 1;
    if (c.wlog > srcLog) c.wlog = srcLog;
    if (c.hlog > c.wlog + 1) c.hlog = c.wlog + 1;
    if (c.clog > c.wlog) c.clog = c.wlog;
    if (c.wlog < 10) c.wlog = 10;
    *out = c;
    return 0;
}

/* ------------------------------------------------------------------ LSB-first bit writer */
typedef struct { uint8_t* p; size_t cap; uint64_t acc; int n; size_t pos; int overflow; } bitw;
static void bw_init(bitw* b, uint8_t* p, size_t cap) { b->p = p; b->cap = cap; b->acc = 0; b->n = 0; b->pos = 0; b->overflow = 0; }
static void bw_add(bitw* b, uint32_t v, int nb)
{
    if (nb == 0) return;
    b->acc |= ((uint64_t)v & ((1ull << nb) - 1)) << b->n;
    b->n += nb;
    while (b->n >= 8) {
        if (b->pos < b->cap) b->p[b->pos] = (uint8_t)b->acc; else b->overflow = 1;
        b->pos++; b->acc >>= 8; b->n -= 8;
    }
}
/* end mark + zero padding; returns total bytes (0 on overflow) */
static size_t bw_close(bitw* b)
{
    bw_add(b, 1, 1);
    if (b->n) { if (b->pos < b->cap) b->p[b->pos] = (uint8_t)b->acc; else b->overflow = 1; b->pos++; }
    return b->overflow ? 0 : b->pos;
}

/* ------------------------------------------------------------------ FSE (compression side) */
typedef struct {
    int log;
    uint16_t cellOf[64 + 1];     /* first entry of each symbol in `next` */
    int16_t  norm[64];
    uint16_t next[4096];         /* per symbol, its cells in table order: value = tableSize + cell */
    unsigned maxSym;
} fse_ctab;

static unsigned fse_optimal_log(unsigned maxLog, size_t n, unsigned maxSym, unsigned minus)
{
    unsigned maxBitsSrc = (unsigned)zo_highbit((uint32_t)(n - 1)) - minus;
    unsigned minBitsSrc = (unsigned)zo_highbit((uint32_t)n) + 1, minBitsSym = (unsigned)zo_highbit(maxSym) + 2;
    unsigned minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    unsigned log = maxLog;
    if (maxBitsSrc < log) log = maxBitsSrc;
    if (minBits > log) log = minBits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

/* secondary distribution method, used when the largest symbol cannot absorb the rounding debt */
static int fse_normalize_m2(int16_t* norm, unsigned log, const unsigned* count, size_t total, unsigned maxSym, int lowProb)
{
    const int16_t UNSET = -2;
    uint32_t distributed = 0;
    uint32_t lowThreshold = (uint32_t)(total >> log);
    uint32_t lowOne = (uint32_t)((total * 3) >> (log + 1));
    for (unsigned s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = (int16_t)lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = UNSET;
    }
    uint32_t toDistribute = (1u << log) - distributed;
    if (toDistribute == 0) return 0;
    if ((total / toDistribute) > lowOne) {
        lowOne = (uint32_t)((total * 3) / (toDistribute * 2));
        for (unsigned s = 0; s <= maxSym; s++)
            if (norm[s] == UNSET && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        toDistribute = (1u << log) - distributed;
    }
    if (distributed == maxSym + 1) {
        unsigned maxV = 0, maxC = 0;
        for (unsigned s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (int16_t)toDistribute;
        return 0;
    }
    if (total == 0) {
        for (unsigned s = 0; toDistribute > 0; s = (s + 1) % (maxSym + 1))
            if (norm[s] > 0) { toDistribute--; norm[s]++; }
        return 0;
    }
    {
        uint64_t vStepLog = 62 - log, mid = (1ull << (vStepLog - 1)) - 1;
        uint64_t rStep = (((1ull << vStepLog) * toDistribute) + mid) / (uint32_t)total;
        uint64_t tmpTotal = mid;
        for (unsigned s = 0; s <= maxSym; s++) if (norm[s] == UNSET) {
            uint64_t end = tmpTotal + (uint64_t)count[s] * rStep;
            uint32_t weight = (uint32_t)(end >> vStepLog) - (uint32_t)(tmpTotal >> vStepLog);
            if (weight < 1) return -ZO_E_GENERIC;
            norm[s] = (int16_t)weight;
            tmpTotal = end;
        }
    }
    return 0;
}

static int fse_normalize(int16_t* norm, unsigned log, const unsigned* count, size_t total, unsigned maxSym, int useLowProb)
{
    static const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    const int lowProb = useLowProb ? -1 : 1;
    const uint64_t scale = 62 - log, step = (1ull << 62) / (uint32_t)total, vStep = 1ull << (scale - 20);
    int still = 1 << log;
    unsigned largest = 0; int16_t largestP = 0;
    uint32_t lowThreshold = (uint32_t)(total >> log);
    for (unsigned s = 0; s <= maxSym; s++) {
        if (count[s] == total) return 0;                    /* caller handles rle before getting here */
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = (int16_t)lowProb; still--; }
        else {
            int16_t proba = (int16_t)(((uint64_t)count[s] * step) >> scale);
            if (proba < 8) {
                uint64_t restToBeat = vStep * rtb[proba];
                proba += ((uint64_t)count[s] * step) - ((uint64_t)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) return fse_normalize_m2(norm, log, count, total, maxSym, lowProb);
    norm[largest] += (int16_t)still;
    return 0;
}

/* distribution header (RFC 8878 4.1.1), returns bytes written */
static size_t fse_write_ncount(uint8_t* out, const int16_t* norm, unsigned maxSym, unsigned log)
{
    bitw b; bw_init(&b, out, 512);
    const unsigned alphabet = maxSym + 1;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbBits = (int)log + 1;
    unsigned sym = 0; int prev0 = 0;
    bw_add(&b, log - 5, 4);
    while (sym < alphabet && remaining > 1) {
        if (prev0) {
            unsigned start = sym;
            while (sym < alphabet && !norm[sym]) sym++;
            if (sym == alphabet) break;
            while (sym >= start + 24) { start += 24; bw_add(&b, 0xFFFF, 16); }
            while (sym >= start + 3) { start += 3; bw_add(&b, 3, 2); }
            bw_add(&b, sym - start, 2);
        }
        {
            int count = norm[sym++];
            int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bw_add(&b, (uint32_t)count, nbBits - (count < max));
            prev0 = (count == 1);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
    }
    if (b.n) { b.p[b.pos++] = (uint8_t)b.acc; }
    return b.pos;
}

/* encoding table: same symbol spread as the decoder's; per symbol the list of its cells in table order */
static void fse_build_ctab(fse_ctab* t, const int16_t* norm, unsigned maxSym, unsigned log)
{
    uint8_t cellSym[4096];
    unsigned size = 1u << log, high = size - 1, step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    t->log = (int)log; t->maxSym = maxSym;
    for (unsigned s = 0; s <= maxSym; s++) { t->norm[s] = norm[s]; if (norm[s] == -1) cellSym[high--] = (uint8_t)s; }
    for (unsigned s = 0; s <= maxSym; s++)
        for (int i = 0; i < norm[s]; i++) {
            cellSym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    unsigned cum = 0;
    for (unsigned s = 0; s <= maxSym; s++) { t->cellOf[s] = (uint16_t)cum; cum += norm[s] == -1 ? 1u : (unsigned)norm[s]; }
    t->cellOf[maxSym + 1] = (uint16_t)cum;
    uint16_t fill[64];
    for (unsigned s = 0; s <= maxSym; s++) fill[s] = t->cellOf[s];
    for (unsigned u = 0; u < size; u++) t->next[fill[cellSym[u]]++] = (uint16_t)(size + u);
}
static void fse_build_rle(fse_ctab* t, unsigned sym)
{
    /* one-cell table: zero bits per symbol (matches FSE_buildCTable_rle, zstd.c:16465) */
    t->log = 0; t->maxSym = sym;
    for (unsigned s = 0; s <= sym; s++) { t->norm[s] = 0; t->cellOf[s] = 0; }
    t->norm[sym] = 1; t->cellOf[sym] = 0; t->cellOf[sym + 1] = 1; t->next[0] = 1;   /* size(1) + cell 0 */
}
/* state value lives in [size, 2*size). The number of bits to flush for symbol s from state v is the unique nb with
 * (v >> nb) in [c, 2c), c = norm[s] (1 for low-prob symbols). */
static inline int fse_nbits(const fse_ctab* t, unsigned s, uint32_t v)
{
    int c = t->norm[s] == -1 ? 1 : t->norm[s];
    if (c == 1) return t->log;
    int maxBits = t->log - zo_highbit((uint32_t)(c - 1));
    return v >= ((uint32_t)c << maxBits) ? maxBits : maxBits - 1;
}
static inline uint32_t fse_first_state(const fse_ctab* t, unsigned s)
{
    /* libzstd starts from a virtual state that flushes the maximum bit count and lands on the symbol's first cells */
    int c = t->norm[s] == -1 ? 1 : t->norm[s];
    if (t->log == 0) return 1;
    int nb = (c > 1) ? t->log - zo_highbit((uint32_t)(c - 1)) : t->log;  /* maxBitsOut */
    uint32_t minStatePlus = (uint32_t)c << nb;
    /* nbBitsOut = (deltaNbBits + 2^15) >> 16 with deltaNbBits = (nb<<16) - minStatePlus */
    uint32_t delta = ((uint32_t)nb << 16) - minStatePlus;
    uint32_t nbOut = (delta + (1u << 15)) >> 16;
    uint32_t value = (nbOut << 16) - delta;
    return t->next[t->cellOf[s] + (value >> nbOut) - (uint32_t)c];
}
static inline uint32_t fse_encode(const fse_ctab* t, bitw* b, uint32_t v, unsigned s)
{
    if (t->log == 0) return v;                           /* rle: nothing to emit */
    int c = t->norm[s] == -1 ? 1 : t->norm[s];
    int nb = fse_nbits(t, s, v);
    bw_add(b, v, nb);
    return t->next[t->cellOf[s] + (v >> nb) - (uint32_t)c];
}

/* ------------------------------------------------------------------ Huffman (compression side) */
typedef struct { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; } hnode;
typedef struct { uint8_t nbBits[256]; uint16_t code[256]; unsigned maxSym; unsigned log; } huf_ctab;

static unsigned huf_bucket(uint32_t c) { return c < 166 ? c : (unsigned)zo_highbit(c) + 158; }

static void huf_insertion(hnode* a, int n)
{
    for (int i = 1; i < n; i++) {
        hnode key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
static int huf_partition(hnode* a, int low, int high)
{
    uint32_t pivot = a[high].count; int i = low - 1;
    for (int j = low; j < high; j++) if (a[j].count > pivot) { i++; hnode t = a[i]; a[i] = a[j]; a[j] = t; }
    hnode t = a[i + 1]; a[i + 1] = a[high]; a[high] = t;
    return i + 1;
}
static void huf_quicksort(hnode* a, int low, int high)
{
    if (high - low < 8) { huf_insertion(a + low, high - low + 1); return; }
    while (low < high) {
        int idx = huf_partition(a, low, high);
        if (idx - low < high - idx) { huf_quicksort(a, low, idx - 1); low = idx + 1; }
        else { huf_quicksort(a, idx + 1, high); high = idx - 1; }
    }
}

/* counts -> length-limited canonical code. Returns the actual maximum code length. */
static unsigned huf_build(huf_ctab* ct, const unsigned* count, unsigned maxSym, unsigned maxBits)
{
    hnode tbl[2 * 256 + 2]; memset(tbl, 0, sizeof tbl);
    hnode* node = tbl + 1;                                  /* node[-1] is the sentinel */
    /* 1. sort by decreasing count: exact buckets below 166 (stable in symbol order), log2 buckets above (quicksorted) */
    { uint16_t base[192 + 1]; uint16_t cur[192 + 1]; memset(base, 0, sizeof base);
      unsigned n1 = maxSym + 1;
      for (unsigned n = 0; n < n1; n++) base[huf_bucket(count[n])]++;
      for (int n = 191; n > 0; n--) { base[n - 1] = (uint16_t)(base[n - 1] + base[n]); }
      /* base[r] = number of symbols whose bucket is >= r, i.e. where bucket r-1's members start */
      for (int n = 0; n <= 192; n++) cur[n] = base[n];
      for (unsigned n = 0; n < n1; n++) {
          unsigned r = huf_bucket(count[n]) + 1;
          unsigned pos = cur[r]++;
          node[pos].count = count[n]; node[pos].byte = (uint8_t)n;
      }
      for (unsigned r = 166; r < 191; r++) {
          int bsize = (int)cur[r] - (int)base[r];
          if (bsize > 1) huf_quicksort(node + base[r], 0, bsize - 1);
      }
    }
    /* 2. two-queue tree construction; leaves are node[0..last], internal nodes start at 256 */
    int last = (int)maxSym;
    while (node[last].count == 0) last--;
    int lowS = last, nodeNb = 256, nodeRoot = nodeNb + lowS - 1, lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (uint16_t)nodeNb;
    nodeNb++; lowS -= 2;
    for (int n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
    node[-1].count = 1u << 31;
    while (nodeNb <= nodeRoot) {
        int n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)nodeNb;
        nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (int n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    for (int n = 0; n <= last; n++) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    /* 3. enforce the length limit (repay the Kraft debt from the cheapest ranks, as libzstd does) */
    unsigned largest = node[last].nbBits;
    if (largest > maxBits) {
        int totalCost = 0; const uint32_t baseCost = 1u << (largest - maxBits);
        int n = last;
        while (node[n].nbBits > maxBits) { totalCost += (int)(baseCost - (1u << (largest - node[n].nbBits))); node[n].nbBits = (uint8_t)maxBits; n--; }
        while (node[n].nbBits == maxBits) --n;
        totalCost >>= (largest - maxBits);
        const uint32_t none = 0xF0F0F0F0u; uint32_t rankLast[14];
        for (int i = 0; i < 14; i++) rankLast[i] = none;
        { unsigned cur = maxBits;
          for (int pos = n; pos >= 0; pos--) { if (node[pos].nbBits >= cur) continue; cur = node[pos].nbBits; rankLast[maxBits - cur] = (uint32_t)pos; } }
        while (totalCost > 0) {
            uint32_t dec = (uint32_t)zo_highbit((uint32_t)totalCost) + 1;
            for (; dec > 1; dec--) {
                uint32_t highPos = rankLast[dec], lowPos = rankLast[dec - 1];
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
        largest = maxBits;
    }
    /* 4. canonical codes: longest rank starts at 0; inside a rank, codes follow symbol order */
    uint16_t perRank[14] = {0}, start[14] = {0};
    for (int n = 0; n <= last; n++) perRank[node[n].nbBits]++;
    { uint16_t min = 0; for (int r = (int)largest; r > 0; r--) { start[r] = min; min = (uint16_t)((min + perRank[r]) >> 1); } }
    memset(ct->nbBits, 0, sizeof ct->nbBits); memset(ct->code, 0, sizeof ct->code);
    for (unsigned n = 0; n <= maxSym; n++) ct->nbBits[node[n].byte] = node[n].nbBits;
    for (unsigned s = 0; s <= maxSym; s++) ct->code[s] = ct->nbBits[s] ? start[ct->nbBits[s]]++ : 0;
    ct->maxSym = maxSym; ct->log = largest;
    return largest;
}

/* FSE-compress the weight list; 0 = not compressible, 1 = single symbol */
static size_t huf_compress_weights(uint8_t* out, const uint8_t* w, size_t n)
{
    unsigned count[13] = {0}, maxSym = 0, maxCount = 0;
    if (n <= 1) return 0;
    for (size_t i = 0; i < n; i++) count[w[i]]++;
    for (unsigned s = 0; s <= 12; s++) { if (count[s]) maxSym = s; if (count[s] > maxCount) maxCount = count[s]; }
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    unsigned log = fse_optimal_log(6, n, maxSym, 2);
    int16_t norm[13];
    if (fse_normalize(norm, log, count, n, maxSym, 0) < 0) return 0;
    size_t h = fse_write_ncount(out, norm, maxSym, log);
    fse_ctab t; fse_build_ctab(&t, norm, maxSym, log);
    if (n <= 2) return 0;
    /* two interleaved states; symbol i belongs to state 1 when i is even. The stream is written last symbol first. */
    bitw b; bw_init(&b, out + h, 512);
    size_t ip = n;
    uint32_t s1, s2;
    if (n & 1) { s1 = fse_first_state(&t, w[--ip]); s2 = fse_first_state(&t, w[--ip]); s1 = fse_encode(&t, &b, s1, w[--ip]); }
    else { s2 = fse_first_state(&t, w[--ip]); s1 = fse_first_state(&t, w[--ip]); }
    while (ip > 0) {
        s2 = fse_encode(&t, &b, s2, w[--ip]);
        s1 = fse_encode(&t, &b, s1, w[--ip]);
    }
    bw_add(&b, s2, (int)log); bw_add(&b, s1, (int)log);
    size_t c = bw_close(&b);
    return c ? h + c : 0;
}

/* tree description: FSE-compressed weights when that is shorter than half the alphabet, else 4-bit weights */
static size_t huf_write_table(uint8_t* out, const huf_ctab* ct)
{
    uint8_t w[256];
    unsigned maxSym = ct->maxSym;
    for (unsigned n = 0; n < maxSym; n++) w[n] = ct->nbBits[n] ? (uint8_t)(ct->log + 1 - ct->nbBits[n]) : 0;
    size_t h = huf_compress_weights(out + 1, w, maxSym);
    if (h > 1 && h < maxSym / 2) { out[0] = (uint8_t)h; return h + 1; }
    if (maxSym > 128) return 0;                            /* cannot be described: caller stores literals raw */
    out[0] = (uint8_t)(128 + (maxSym - 1));
    w[maxSym] = 0;
    for (unsigned n = 0; n < maxSym; n += 2) out[n / 2 + 1] = (uint8_t)((w[n] << 4) + w[n + 1]);
    return (maxSym + 1) / 2 + 1;
}

static size_t huf_encode_1x(uint8_t* out, size_t cap, const uint8_t* src, size_t n, const huf_ctab* ct)
{
    bitw b; bw_init(&b, out, cap);
    for (size_t i = n; i-- > 0;) bw_add(&b, ct->code[src[i]], ct->nbBits[src[i]]);
    return bw_close(&b);
}
static size_t huf_encode_4x(uint8_t* out, size_t cap, const uint8_t* src, size_t n, const huf_ctab* ct)
{
    size_t seg = (n + 3) / 4, pos = 6;
    if (n < 12 || cap < 6 + 1 + 1 + 1 + 8) return 0;
    for (int k = 0; k < 4; k++) {
        size_t len = k < 3 ? seg : n - 3 * seg;
        size_t c = huf_encode_1x(out + pos, cap - pos, src + k * seg, len, ct);
        if (c == 0 || c > 65535) return 0;
        if (k < 3) zo_wr16(out + 2 * k, (uint16_t)c);
        pos += c;
    }
    return pos;
}

/* literals section. Returns bytes written. */
static size_t zo_raw_literals(uint8_t* out, const uint8_t* lit, size_t n, unsigned type, int rle)
{
    unsigned fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) out[0] = (uint8_t)(type + (n << 3));
    else if (fl == 2) zo_wr16(out, (uint16_t)(type + (1 << 2) + (n << 4)));
    else zo_wr32(out, (uint32_t)(type + (3 << 2) + (n << 4)));
    if (rle) { out[fl] = lit[0]; return fl + 1; }
    memcpy(out + fl, lit, n);
    return fl + n;
}

/* entropy tables inherited from a dictionary (the only "previous block" a single-block frame can have):
 * repeat modes follow libzstd: 0 none, 1 check (usable if it covers the data), 2 valid (zstd.c:3097-3101, FSE_repeat) */
typedef struct { huf_ctab huf; int hufRepeat; fse_ctab ll, of, ml; int llRepeat, ofRepeat, mlRepeat; } zo_entropy;

static size_t huf_estimate(const huf_ctab* ct, const unsigned* count, unsigned maxSym)
{
    size_t bits = 0;
    for (unsigned s = 0; s <= maxSym; s++) bits += (size_t)ct->nbBits[s] * count[s];
    return bits >> 3;
}
static int huf_validate(const huf_ctab* ct, const unsigned* count, unsigned maxSym)
{
    if (ct->maxSym < maxSym) return 0;
    for (unsigned s = 0; s <= maxSym; s++) if (count[s] && !ct->nbBits[s]) return 0;
    return 1;
}

/* *usedNew = 1 and *newTab = the table when the section is emitted with a freshly built Huffman table (the table the NEXT block may
 * reuse after validation, zstd.c:21046-21049); otherwise the previous table stays current. */
static size_t zo_compress_literals(uint8_t* out, size_t cap, const uint8_t* lit, size_t n, size_t nbSeq, const zo_entropy* prev,
                                   huf_ctab* newTab, int* usedNew)
{
    if (usedNew) *usedNew = 0;
    if (prev && prev->hufRepeat) {
        /* ZSTD_compressLiterals + HUF_compress_internal with a candidate previous table (zstd.c:20932, :18089) */
        const size_t lh = 3 + (n >= 1024) + (n >= 16384);
        int single = n < 256;
        int repeat = prev->hufRepeat;
        const size_t minLits = repeat == 2 ? 6 : 64;
        if (n < minLits) return zo_raw_literals(out, lit, n, 0, 0);
        const int suspect = (nbSeq == 0) || (n / nbSeq >= 20);
        const int preferRepeat = n <= 1024;
        if (repeat == 2 && lh == 3) single = 1;
        uint8_t* body = out + lh;
        const size_t bcap = cap - lh;
        size_t cl = 0; int useOld = 0, done = 0, rle = 0;
        huf_ctab nt; size_t h = 0;
        unsigned count[256]; unsigned maxSym = 0, largest = 0;
        if (preferRepeat && repeat == 2) { useOld = 1; }
        else {
            if (suspect && n >= 4096 * 10) {
                unsigned a = 0, b = 0; unsigned c[256];
                memset(c, 0, sizeof c); for (size_t i = 0; i < 4096; i++) c[lit[i]]++;
                for (int s2 = 0; s2 < 256; s2++) if (c[s2] > a) a = c[s2];
                memset(c, 0, sizeof c); for (size_t i = n - 4096; i < n; i++) c[lit[i]]++;
                for (int s2 = 0; s2 < 256; s2++) if (c[s2] > b) b = c[s2];
                if (a + b <= ((2 * 4096) >> 7) + 4) { done = 1; cl = 0; }
            }
            if (!done) {
                memset(count, 0, sizeof count);
                for (size_t i = 0; i < n; i++) count[lit[i]]++;
                for (unsigned s2 = 0; s2 < 256; s2++) { if (count[s2]) maxSym = s2; if (count[s2] > largest) largest = count[s2]; }
                if (largest == n) { done = 1; rle = 1; }
                else if (largest <= (n >> 7) + 4) { done = 1; cl = 0; }
            }
            if (!done) {
                if (repeat == 1 && !huf_validate(&prev->huf, count, maxSym)) repeat = 0;
                if (preferRepeat && repeat != 0) useOld = 1;
                else {
                    unsigned log = fse_optimal_log(11, n, maxSym, 1);
                    huf_build(&nt, count, maxSym, log);
                    h = huf_write_table(body, &nt);
                    if (h == 0) { done = 1; cl = 0; }       /* table cannot be described: error -> raw literals */
                    else if (repeat != 0) {
                        size_t oldSize = huf_estimate(&prev->huf, count, maxSym), newSize = huf_estimate(&nt, count, maxSym);
                        if (oldSize <= h + newSize || h + 12 >= n) useOld = 1;
                    }
                    if (!done && !useOld) {
                        if (h + 12 >= n) { done = 1; cl = 0; }
                        else {
                            repeat = 0;
                            size_t c = single ? huf_encode_1x(body + h, bcap - h, lit, n, &nt) : huf_encode_4x(body + h, bcap - h, lit, n, &nt);
                            cl = c ? h + c : 0; if (cl >= n - 1) cl = 0;
                            done = 1;
                        }
                    }
                }
            }
        }
        if (!done && useOld) {
            size_t c = single ? huf_encode_1x(body, bcap, lit, n, &prev->huf) : huf_encode_4x(body, bcap, lit, n, &prev->huf);
            cl = c; if (cl >= n - 1) cl = 0;
        }
        if (rle) return zo_raw_literals(out, lit, n, 1, 1);
        if (cl == 0 || cl >= n - ((n >> 6) + 2)) return zo_raw_literals(out, lit, n, 0, 0);
        if (cl == 1) {
            int same = 1; for (size_t i = 1; i < n; i++) if (lit[i] != lit[0]) { same = 0; break; }
            if (n >= 8 || same) return zo_raw_literals(out, lit, n, 1, 1);
        }
        const unsigned hType = repeat != 0 ? 3u : 2u;
        if (hType == 2 && usedNew) { *usedNew = 1; *newTab = nt; }
        if (lh == 3) { uint32_t v = hType + ((uint32_t)(!single) << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 14); zo_wr24(out, v); }
        else if (lh == 4) { zo_wr32(out, hType + (2u << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 18)); }
        else { zo_wr32(out, hType + (3u << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 22)); out[4] = (uint8_t)(cl >> 10); }
        return lh + cl;
    }

    const size_t lh = 3 + (n >= 1024) + (n >= 16384);
    const int single = n < 256;
    if (n < 64) return zo_raw_literals(out, lit, n, 0, 0);           /* dfast, no previous table: 8 << 3 */
    const int suspect = (nbSeq == 0) || (n / nbSeq >= 20);
    unsigned count[256]; unsigned maxSym = 0; unsigned largest = 0;
    if (suspect && n >= 4096 * 10) {
        unsigned a = 0, b = 0; unsigned c[256];
        memset(c, 0, sizeof c); for (size_t i = 0; i < 4096; i++) c[lit[i]]++;
        for (int s = 0; s < 256; s++) if (c[s] > a) a = c[s];
        memset(c, 0, sizeof c); for (size_t i = n - 4096; i < n; i++) c[lit[i]]++;
        for (int s = 0; s < 256; s++) if (c[s] > b) b = c[s];
        if (a + b <= ((2 * 4096) >> 7) + 4) return zo_raw_literals(out, lit, n, 0, 0);
    }
    memset(count, 0, sizeof count);
    for (size_t i = 0; i < n; i++) count[lit[i]]++;
    for (unsigned s = 0; s < 256; s++) { if (count[s]) maxSym = s; if (count[s] > largest) largest = count[s]; }
    if (largest == n) return zo_raw_literals(out, lit, n, 1, 1);     /* one symbol: RLE literals */
    if (largest <= (n >> 7) + 4) return zo_raw_literals(out, lit, n, 0, 0);
    huf_ctab ct;
    unsigned log = fse_optimal_log(11, n, maxSym, 1);
    huf_build(&ct, count, maxSym, log);
    uint8_t* body = out + lh;
    size_t h = huf_write_table(body, &ct);
    if (h == 0 || h + 12 >= n) return zo_raw_literals(out, lit, n, 0, 0);
    size_t c = single ? huf_encode_1x(body + h, cap - lh - h, lit, n, &ct) : huf_encode_4x(body + h, cap - lh - h, lit, n, &ct);
    size_t cl = c ? h + c : 0;
    if (cl >= n - 1) cl = 0;
    if (cl == 0 || cl >= n - ((n >> 6) + 2)) return zo_raw_literals(out, lit, n, 0, 0);
    if (usedNew) { *usedNew = 1; *newTab = ct; }
    /* header: type 2 (compressed), size format by header length */
    if (lh == 3) { uint32_t v = 2 + ((uint32_t)(!single) << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 14); zo_wr24(out, v); }
    else if (lh == 4) { zo_wr32(out, 2 + (2u << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 18)); }
    else { zo_wr32(out, 2 + (3u << 2) + ((uint32_t)n << 4) + ((uint32_t)cl << 22)); out[4] = (uint8_t)(cl >> 10); }
    return lh + cl;
}

/* ------------------------------------------------------------------ double-fast match finder */
typedef struct { uint32_t offBase; uint32_t litLength; uint32_t matchLength; } zo_seq;

static inline uint32_t hash_n(const uint8_t* p, int hbits, int mls)
{
    uint64_t u = zo_rd64(p);
    switch (mls) {
    case 5: return (uint32_t)(((u << 24) * 889523592379ull) >> (64 - hbits));
    case 6: return (uint32_t)(((u << 16) * 227718039650203ull) >> (64 - hbits));
    case 7: return (uint32_t)(((u << 8) * 58295818150454627ull) >> (64 - hbits));
    case 8: return (uint32_t)((u * 0xCF1BBCDCB7A56463ull) >> (64 - hbits));
    default: return (uint32_t)(((uint32_t)u * 2654435761u) >> (32 - hbits));
    }
}
static size_t common_len(const uint8_t* a, const uint8_t* b, const uint8_t* aend)
{
    const uint8_t* s = a;
    while (a + 8 <= aend) { uint64_t d = zo_rd64(a) ^ zo_rd64(b); if (d) return (size_t)(a - s) + (size_t)(__builtin_ctzll(d) >> 3); a += 8; b += 8; }
    while (a < aend && *a == *b) { a++; b++; }
    return (size_t)(a - s);
}

/* One block that is also the whole frame (no history). Positions are frame-relative; table cells hold pos+2 so that
 * 0 means "empty" and the reference's index comparisons (>= lowest for matches at ip, > lowest for the long match
 * at ip+1) keep their meaning with lowest == 2. Returns the number of sequences. */
/* `frame` is the first byte of the frame (index 2), `src` the block; rep[] carries the two repcodes in and out
 * (zstd.c:31091-31098 and :31175-31182: offsets larger than the history are parked and restored). Tables are the caller's:
 * zeroed before the first block, kept between blocks. */
static size_t zo_dfast_g(zo_seq* seqs, uint8_t* lits, size_t* litSize, const uint8_t* frame, const uint8_t* src, size_t srcSize, const zo_cpar* cp,
                         uint32_t* hashLong, uint32_t* hashSmall, uint32_t rep[2])
{
    const int hl = cp->hlog, hs = cp->clog;
    const int mls = cp->mml <= 4 ? 4 : cp->mml >= 7 ? 7 : cp->mml;
    const uint8_t* const base = frame - 2;                     /* index = frame position + 2 */
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    /* lowest index a match may start at: the frame start, or what the window still covers at the END of this block
     * (ZSTD_getLowestPrefixIndex, zstd.c:20566, after ZSTD_window_enforceMaxDist :20386) */
    const uint32_t maxDist = 1u << cp->wlog;
    const uint32_t endIndex = (uint32_t)(iend - base);
    const uint32_t LOW = endIndex - 2 > maxDist ? endIndex - maxDist : 2;
    const uint8_t* anchor = src;
    const uint8_t* ip = src + ((uint32_t)(src - base) == LOW); /* an empty prefix skips its first position (zstd.c:31091): the frame's first block,
                                                                * or a later block whose window reaches back exactly to its own start */
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    { const uint32_t curr = (uint32_t)(ip - base); uint32_t maxRep = curr - 2 > maxDist ? maxDist : curr - 2;
      if (off2 > maxRep) { saved2 = off2; off2 = 0; } if (off1 > maxRep) { saved1 = off1; off1 = 0; } }
    size_t nseq = 0; uint8_t* lp = lits;
    if (srcSize < 8) { memcpy(lp, src, srcSize); *litSize = srcSize; return 0; }
#define STORE(LL, OFFBASE, ML) do { size_t ll_ = (LL); memcpy(lp, anchor, ll_); lp += ll_; \
        seqs[nseq].litLength = (uint32_t)ll_; seqs[nseq].offBase = (OFFBASE); seqs[nseq].matchLength = (uint32_t)(ML); nseq++; } while (0)
    for (;;) {
        size_t step = 1; const uint8_t* nextStep = ip + 256; const uint8_t* ip1 = ip + step;
        size_t mLength; uint32_t offset, curr = 0;
        if (ip1 > ilimit) break;
        uint32_t hl0 = hash_n(ip, hl, 8), idxl0 = hashLong[hl0];
        uint32_t hl1 = 0, idxl1 = 0;
        int found = 0;      /* 1: repcode at ip+1, 2: offset match */
        do {
            uint32_t hs0 = hash_n(ip, hs, mls), idxs0 = hashSmall[hs0];
            curr = (uint32_t)(ip - base);
            hashLong[hl0] = hashSmall[hs0] = curr;
            if (off1 > 0 && zo_rd32(ip + 1 - off1) == zo_rd32(ip + 1)) {
                mLength = common_len(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                STORE(ip - anchor, 1, mLength);
                found = 1; break;
            }
            hl1 = hash_n(ip1, hl, 8);
            if (idxl0 >= LOW && zo_rd64(base + idxl0) == zo_rd64(ip)) {
                const uint8_t* m = base + idxl0;
                mLength = common_len(ip + 8, m + 8, iend) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > base + LOW && ip[-1] == m[-1]) { ip--; m--; mLength++; }
                found = 2; break;
            }
            idxl1 = hashLong[hl1];
            if (idxs0 >= LOW && zo_rd32(base + idxs0) == zo_rd32(ip)) {
                const uint8_t* m = base + idxs0;
                mLength = common_len(ip + 4, m + 4, iend) + 4;
                offset = (uint32_t)(ip - m);
                if (idxl1 > LOW && zo_rd64(base + idxl1) == zo_rd64(ip1)) {
                    const uint8_t* m1 = base + idxl1;
                    size_t l1 = common_len(ip1 + 8, m1 + 8, iend) + 8;
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
            STORE(ip - anchor, offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            uint32_t ins = curr + 2;
            hashLong[hash_n(base + ins, hl, 8)] = ins;
            hashLong[hash_n(ip - 2, hl, 8)] = (uint32_t)(ip - 2 - base);
            hashSmall[hash_n(base + ins, hs, mls)] = ins;
            hashSmall[hash_n(ip - 1, hs, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit && off2 > 0 && zo_rd32(ip) == zo_rd32(ip - off2)) {
                size_t r = common_len(ip + 4, ip + 4 - off2, iend) + 4;
                uint32_t t = off2; off2 = off1; off1 = t;
                hashSmall[hash_n(ip, hs, mls)] = (uint32_t)(ip - base);
                hashLong[hash_n(ip, hl, 8)] = (uint32_t)(ip - base);
                STORE(0, 1, r);
                ip += r; anchor = ip;
            }
        }
    }
#undef STORE
    { size_t last = (size_t)(iend - anchor); memcpy(lp, anchor, last); lp += last; }
    *litSize = (size_t)(lp - lits);
    if (saved1 != 0 && off1 != 0) saved2 = saved1;
    rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
    return nseq;
}
static size_t zo_dfast(zo_seq* seqs, uint8_t* lits, size_t* litSize, const uint8_t* src, size_t srcSize, const zo_cpar* cp,
                       uint32_t* hashLong, uint32_t* hashSmall)
{
    uint32_t rep[2] = {1, 4};
    memset(hashLong, 0, sizeof(uint32_t) << cp->hlog);
    memset(hashSmall, 0, sizeof(uint32_t) << cp->clog);
    return zo_dfast_g(seqs, lits, litSize, src, src, srcSize, cp, hashLong, hashSmall, rep);
}


/* ------------------------------------------------------------------ fast strategy (levels 1-2 and negative levels)
 * Restates ZSTD_compressBlock_fast_noDict_generic (zstd.c:31906) for a block that is the whole frame. One hash table (hashLog bits,
 * minMatch bytes hashed), cells hold position + 2 (0 = empty). Positions are examined in pairs (p, p+1) `step` apart; the pair
 * distance grows by one for every 128 bytes advanced without a match; a repcode test sits two positions ahead of the first of a
 * pair; the table write for the second position of a pair is skipped after a hit only when step > 4. */
static size_t zo_fast_g(zo_seq* seqs, uint8_t* lits, size_t* litSize, const uint8_t* frame, const uint8_t* src, size_t srcSize, const zo_cpar* cp,
                        uint32_t* table, uint32_t rep[2])
{
    const int hlog = cp->hlog;
    const int mls = cp->mml <= 4 ? 4 : cp->mml >= 7 ? 7 : cp->mml;
    const size_t stepSize = (size_t)cp->tlen + !cp->tlen + 1;
    const uint8_t* const base = frame - 2;
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    const uint32_t maxDist = 1u << cp->wlog;
    const uint32_t endIndex = (uint32_t)(iend - base);
    const uint32_t LOW = endIndex - 2 > maxDist ? endIndex - maxDist : 2;
    const uint8_t* anchor = src;
    const uint8_t* ip0 = src + ((uint32_t)(src - base) == LOW);   /* empty prefix (zstd.c:31958), as in zo_dfast */
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    { const uint32_t curr = (uint32_t)(ip0 - base); uint32_t maxRep = curr - 2 > maxDist ? maxDist : curr - 2;
      if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; } if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; } }
    size_t nseq = 0; uint8_t* lp = lits;
#define STORE(LL, OFFBASE, ML) do { size_t ll_ = (LL); memcpy(lp, anchor, ll_); lp += ll_; \
        seqs[nseq].litLength = (uint32_t)ll_; seqs[nseq].offBase = (OFFBASE); seqs[nseq].matchLength = (uint32_t)(ML); nseq++; } while (0)
#define IDX(p) ((uint32_t)((p) - base))
    if (srcSize >= 8) for (;;) {
        size_t step = stepSize;
        const uint8_t* nextStep = ip0 + 128;
        const uint8_t* ip1 = ip0 + 1; const uint8_t* ip2 = ip0 + step; const uint8_t* ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        uint32_t hash0 = hash_n(ip0, hlog, mls), hash1 = hash_n(ip1, hlog, mls);
        uint32_t matchIdx = table[hash0];
        uint32_t current0 = 0, offBase = 0; size_t mLength = 0; const uint8_t* match0 = NULL;
        int found = 0;      /* 1 repcode, 2 offset */
        do {
            const uint32_t rval = zo_rd32(ip2 - rep1);
            current0 = IDX(ip0); table[hash0] = current0;
            if (zo_rd32(ip2) == rval && rep1 > 0) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = (ip0[-1] == match0[-1]);
                ip0 -= mLength; match0 -= mLength;
                offBase = 1; mLength += 4;
                table[hash1] = IDX(ip1);
                found = 1; break;
            }
            if (matchIdx >= LOW && zo_rd32(base + matchIdx) == zo_rd32(ip0)) { table[hash1] = IDX(ip1); found = 2; break; }
            matchIdx = table[hash1];
            hash0 = hash1; hash1 = hash_n(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = IDX(ip0); table[hash0] = current0;
            if (matchIdx >= LOW && zo_rd32(base + matchIdx) == zo_rd32(ip0)) { if (step <= 4) table[hash1] = IDX(ip1); found = 2; break; }
            matchIdx = table[hash1];
            hash0 = hash1; hash1 = hash_n(ip2, hlog, mls);
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
        mLength += common_len(ip0 + mLength, match0 + mLength, iend);
        STORE(ip0 - anchor, offBase, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {
            table[hash_n(base + current0 + 2, hlog, mls)] = current0 + 2;
            table[hash_n(ip0 - 2, hlog, mls)] = IDX(ip0 - 2);
            if (rep2 > 0) {
                while (ip0 <= ilimit && zo_rd32(ip0) == zo_rd32(ip0 - rep2)) {
                    const size_t rLength = common_len(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    uint32_t t = rep2; rep2 = rep1; rep1 = t;
                    table[hash_n(ip0, hlog, mls)] = IDX(ip0);
                    ip0 += rLength;
                    STORE(0, 1, rLength);
                    anchor = ip0;
                }
            }
        }
    }
#undef IDX
#undef STORE
    { size_t last = (size_t)(iend - anchor); memcpy(lp, anchor, last); lp += last; }
    *litSize = (size_t)(lp - lits);
    if (saved1 != 0 && rep1 != 0) saved2 = saved1;
    rep[0] = rep1 ? rep1 : saved1; rep[1] = rep2 ? rep2 : saved2;
    return nseq;
}
static size_t zo_fast(zo_seq* seqs, uint8_t* lits, size_t* litSize, const uint8_t* src, size_t srcSize, const zo_cpar* cp, uint32_t* table)
{
    uint32_t rep[2] = {1, 4};
    memset(table, 0, sizeof(uint32_t) << cp->hlog);
    return zo_fast_g(seqs, lits, litSize, src, src, srcSize, cp, table, rep);
}

/* ------------------------------------------------------------------ dictionary (attached CDict, sources <= 16 KiB) */
/* Restates: ZSTD_createCDict_advanced2 (zstd.c:28614) parameters, ZSTD_loadCEntropy (:28015), ZSTD_loadDictionaryContent (:27895),
 * ZSTD_fillDoubleHashTableForCDict (:30952, tagged cells :20636), ZSTD_resetCCtx_byAttachingCDict (:25279) and
 * ZSTD_compressBlock_doubleFast_dictMatchState_generic (:31262). Index space: dictionary content byte k has index 2 + k, the
 * source starts right after it (index CE = 2 + contentSize), so offsets are plain index differences. */
typedef struct {
    zo_cpar cp;                                  /* parameters the CDict tables were built with */
    uint32_t* hashLong; uint32_t* hashSmall;     /* cell = index << 8 | tag */
    const uint8_t* content; size_t contentSize;
    uint32_t dictID;
    uint32_t rep[3];
    zo_entropy ent;                              /* hufRepeat == 0 and *Repeat == 0 for raw-content dictionaries */
} zo_cdict;

static void zo_cdict_free(zo_cdict* d) { if (d) { free(d->hashLong); free(d->hashSmall); free(d); } }

static int zo_cdict_params(zo_cpar* out, int level, size_t dictSize)
{
    if (level == 0) level = 3;
    if (level < 1 || level > 4) return -ZO_E_PARAM_UNSUPPORTED;
    uint64_t rSize = (uint64_t)dictSize + 499;                  /* unknown source size: "size hint" wraps to dictSize + 499 */
    unsigned tableID = (rSize <= 256u * 1024) + (rSize <= 128u * 1024) + (rSize <= 16u * 1024);
    zo_cpar c = zo_rows[tableID][level];
    const uint64_t srcSize = 513;                               /* createCDict mode assumes a small source */
    uint32_t t = (uint32_t)(srcSize + dictSize);
    int srcLog = (t < 64) ? 6 : zo_highbit(t - 1) + 1;
    if (c.wlog > srcLog) c.wlog = srcLog;
    {   uint64_t windowSize = 1ull << c.wlog;
        int dawl = c.wlog;
        if (windowSize < dictSize + srcSize) { uint64_t dw = dictSize + windowSize; dawl = dw >= (1ull << 31) ? 31 : zo_highbit((uint32_t)dw - 1) + 1; }
        if (c.hlog > dawl + 1) c.hlog = dawl + 1;
        if (c.clog > dawl) c.clog = dawl;
    }
    if (c.wlog < 10) c.wlog = 10;
    if (c.hlog > 24) c.hlog = 24;
    if (c.clog > 24) c.clog = 24;
    *out = c;
    return 0;
}

static void huf_ctab_from_weights(huf_ctab* ct, const uint8_t* w, unsigned count, unsigned log)
{
    /* HUF_readCTable (zstd.c:17048): lengths from weights, canonical values per rank in symbol order */
    uint16_t perRank[16] = {0}, start[16] = {0};
    memset(ct->nbBits, 0, sizeof ct->nbBits); memset(ct->code, 0, sizeof ct->code);
    for (unsigned s = 0; s < count; s++) { ct->nbBits[s] = w[s] ? (uint8_t)(log + 1 - w[s]) : 0; perRank[ct->nbBits[s]]++; }
    { uint16_t min = 0; for (int r = (int)log; r > 0; r--) { start[r] = min; min = (uint16_t)((min + perRank[r]) >> 1); } }
    for (unsigned s = 0; s < count; s++) ct->code[s] = ct->nbBits[s] ? start[ct->nbBits[s]]++ : 0;
    ct->maxSym = count - 1; ct->log = log;
}

static int ncount_repeat(const int16_t* norm, unsigned dictMax, unsigned needMax)
{
    if (dictMax < needMax) return 1;
    for (unsigned s = 0; s <= needMax; s++) if (norm[s] == 0) return 1;
    return 2;
}

static zo_cdict* zo_cdict_create(const uint8_t* dict, size_t dictSize, int level, int* err)
{
    zo_cdict* d = (zo_cdict*)calloc(1, sizeof(zo_cdict));
    if (!d) { *err = -ZO_E_MEMORY; return NULL; }
    *err = zo_cdict_params(&d->cp, level, dictSize);
    if (*err < 0 || d->cp.strat != 2) { if (*err >= 0) *err = -ZO_E_PARAM_UNSUPPORTED; free(d); return NULL; }
    d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
    const uint8_t* p = dict; const uint8_t* end = dict + dictSize;
    if (dictSize >= 8 && zo_rd32(dict) == ZO_DICT_MAGIC) {
        d->dictID = zo_rd32(dict + 4);
        p += 8;
        uint8_t w[256]; unsigned cnt, log;
        int r = zo_huf_read_weights(w, &cnt, &log, p, (size_t)(end - p));
        if (r < 0 || log > 12) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; }
        huf_ctab_from_weights(&d->ent.huf, w, cnt, log);
        { int zero = 0; for (unsigned s2 = 0; s2 < cnt; s2++) zero |= (w[s2] == 0); d->ent.hufRepeat = (!zero && cnt == 256) ? 2 : 1; }
        p += r;
        int16_t ofN[64], mlN[64], llN[64]; unsigned ofMax = ZO_MAXOFF, mlMax = ZO_MAXML, llMax = ZO_MAXLL, ofLog, mlLog, llLog;
        memset(ofN, 0, sizeof ofN); memset(mlN, 0, sizeof mlN); memset(llN, 0, sizeof llN);
        r = zo_fse_read_ncount(ofN, &ofMax, &ofLog, p, (size_t)(end - p)); if (r < 0 || ofLog > 8) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; } p += r;
        fse_build_ctab(&d->ent.of, ofN, ZO_MAXOFF, ofLog);              /* all offset symbols, like the reference */
        r = zo_fse_read_ncount(mlN, &mlMax, &mlLog, p, (size_t)(end - p)); if (r < 0 || mlLog > 9) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; } p += r;
        fse_build_ctab(&d->ent.ml, mlN, mlMax, mlLog);
        d->ent.mlRepeat = ncount_repeat(mlN, mlMax, ZO_MAXML);
        r = zo_fse_read_ncount(llN, &llMax, &llLog, p, (size_t)(end - p)); if (r < 0 || llLog > 9) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; } p += r;
        fse_build_ctab(&d->ent.ll, llN, llMax, llLog);
        d->ent.llRepeat = ncount_repeat(llN, llMax, ZO_MAXLL);
        if (p + 12 > end) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; }
        for (int i = 0; i < 3; i++) d->rep[i] = zo_rd32(p + 4 * i);
        p += 12;
        size_t cs = (size_t)(end - p);
        { unsigned offcodeMax = ZO_MAXOFF; uint32_t maxOffset = (uint32_t)cs + 128 * 1024; offcodeMax = (unsigned)zo_highbit(maxOffset);
          d->ent.ofRepeat = ncount_repeat(ofN, ofMax, offcodeMax < ZO_MAXOFF ? offcodeMax : ZO_MAXOFF); }
        for (int i = 0; i < 3; i++) if (d->rep[i] == 0 || d->rep[i] > cs) { *err = -ZO_E_DICT_CORRUPTED; free(d); return NULL; }
    }
    d->content = p; d->contentSize = (size_t)(end - p);
    d->hashLong = (uint32_t*)calloc((size_t)1 << d->cp.hlog, 4);
    d->hashSmall = (uint32_t*)calloc((size_t)1 << d->cp.clog, 4);
    if (!d->hashLong || !d->hashSmall) { *err = -ZO_E_MEMORY; zo_cdict_free(d); return NULL; }
    /* index only the tail the tables can reasonably address; every third position into both tables, the two in between
     * into the long table when its cell is still empty */
    {   size_t cs = d->contentSize, startOff = 0;
        int mx = d->cp.hlog + 3 > d->cp.clog + 1 ? d->cp.hlog + 3 : d->cp.clog + 1; if (mx > 31) mx = 31;
        size_t maxDict = (size_t)1 << mx;
        if (cs > maxDict) startOff = cs - maxDict;
        if (cs - startOff > 8) {
            const uint8_t* base = d->content - 2;
            const int mls = d->cp.mml <= 4 ? 4 : d->cp.mml >= 7 ? 7 : d->cp.mml;
            const uint8_t* ip = d->content + startOff; const uint8_t* iend = d->content + cs - 8;
            for (; ip + 2 <= iend; ip += 3) {
                uint32_t curr = (uint32_t)(ip - base);
                for (uint32_t i = 0; i < 3; i++) {
                    uint32_t sm = hash_n(ip + i, d->cp.clog + 8, mls), lg = hash_n(ip + i, d->cp.hlog + 8, 8);
                    if (i == 0) d->hashSmall[sm >> 8] = ((curr + i) << 8) | (sm & 255);
                    if (i == 0 || d->hashLong[lg >> 8] == 0) d->hashLong[lg >> 8] = ((curr + i) << 8) | (lg & 255);
                }
            }
        }
    }
    return d;
}

/* byte at unified index i (dictionary content below CE, source at and above it) */
typedef struct { const uint8_t* content; const uint8_t* src; uint32_t CE; uint32_t end; } zo_space;
static inline uint8_t sp_byte(const zo_space* sp, uint32_t i) { return i < sp->CE ? sp->content[i - 2] : sp->src[i - sp->CE]; }
static inline uint32_t sp_rd32(const zo_space* sp, uint32_t i)
{
    if (i >= sp->CE) return zo_rd32(sp->src + (i - sp->CE));
    if (i + 4 <= sp->CE) return zo_rd32(sp->content + (i - 2));
    uint32_t v = 0; for (int k = 0; k < 4; k++) v |= (uint32_t)sp_byte(sp, i + k) << (8 * k); return v;
}
static inline uint64_t sp_rd64(const zo_space* sp, uint32_t i)
{
    if (i >= sp->CE) return zo_rd64(sp->src + (i - sp->CE));
    if (i + 8 <= sp->CE) return zo_rd64(sp->content + (i - 2));
    uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)sp_byte(sp, i + k) << (8 * k); return v;
}
/* common length of source position ip (index) and match index m, the match continuing from the dictionary end into the
 * source start (ZSTD_count_2segments, zstd.c:20034) */
static uint32_t sp_count(const zo_space* sp, uint32_t ip, uint32_t m)
{
    uint32_t n = 0;
    while (ip + n < sp->end && sp_byte(sp, ip + n) == sp_byte(sp, m + n)) n++;
    return n;
}

static size_t zo_dfast_dict(zo_seq* seqs, uint8_t* lits, size_t* litSize, const uint8_t* src, size_t srcSize, const zo_cpar* cp,
                            const zo_cdict* d, uint32_t* hashLong, uint32_t* hashSmall)
{
    const int hl = cp->hlog, hs = cp->clog;
    const int mls = cp->mml <= 4 ? 4 : cp->mml >= 7 ? 7 : cp->mml;
    const uint32_t CE = 2 + (uint32_t)d->contentSize;          /* prefixLowestIndex == first source index */
    const uint32_t dictStart = 2;
    zo_space sp = { d->content, src, CE, CE + (uint32_t)srcSize };
    const int dhl = d->cp.hlog + 8, dhs = d->cp.clog + 8;
    const uint32_t iend = CE + (uint32_t)srcSize;
    uint32_t ip = CE, anchor = CE;
    uint32_t off1 = d->rep[0], off2 = d->rep[1];
    size_t nseq = 0; uint8_t* lp = lits;
    memset(hashLong, 0, sizeof(uint32_t) << hl);
    memset(hashSmall, 0, sizeof(uint32_t) << hs);
    if (srcSize < 8) { memcpy(lp, src, srcSize); *litSize = srcSize; return 0; }
    const uint32_t ilimit = iend - 8;
#define SRC(i) (src + ((i) - CE))
#define STORE(LL, OFFBASE, ML) do { size_t ll_ = (LL); memcpy(lp, SRC(anchor), ll_); lp += ll_; \
        seqs[nseq].litLength = (uint32_t)ll_; seqs[nseq].offBase = (OFFBASE); seqs[nseq].matchLength = (uint32_t)(ML); nseq++; } while (0)
    while (ip < ilimit) {
        uint32_t mLength = 0, offset = 0;
        const uint32_t h2 = hash_n(SRC(ip), hl, 8), h = hash_n(SRC(ip), hs, mls);
        const uint32_t dTagL = hash_n(SRC(ip), dhl, 8), dTagS = hash_n(SRC(ip), dhs, mls);
        const uint32_t dEntL = d->hashLong[dTagL >> 8], dEntS = d->hashSmall[dTagS >> 8];
        const int tagL = (dEntL & 255) == (dTagL & 255), tagS = (dEntS & 255) == (dTagS & 255);
        const uint32_t curr = ip;
        const uint32_t mIdxL = hashLong[h2]; uint32_t mIdxS = hashSmall[h];
        const uint32_t repIndex = curr + 1 - off1;
        hashLong[h2] = hashSmall[h] = curr;
        int found = 0;          /* 1 repcode stored, 2 offset match */
        if (((uint32_t)((CE - 1) - repIndex) >= 3) && sp_rd32(&sp, repIndex) == zo_rd32(SRC(ip + 1))) {
            mLength = sp_count(&sp, ip + 1 + 4, repIndex + 4) + 4;
            ip++;
            STORE(ip - anchor, 1, mLength);
            found = 1;
        } else {
            int shortCand = 0; uint32_t match = 0;
            if (mIdxL >= CE && zo_rd64(SRC(mIdxL)) == zo_rd64(SRC(ip))) {
                uint32_t m = mIdxL;
                mLength = sp_count(&sp, ip + 8, m + 8) + 8;
                offset = ip - m;
                while (ip > anchor && m > CE && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; }
                found = 2;
            } else if (tagL) {
                uint32_t m = dEntL >> 8;
                if (m > dictStart && sp_rd64(&sp, m) == zo_rd64(SRC(ip))) {
                    mLength = sp_count(&sp, ip + 8, m + 8) + 8;
                    offset = curr - m;
                    while (ip > anchor && m > dictStart && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; }
                    found = 2;
                }
            }
            if (!found) {
                if (mIdxS > CE) { if (zo_rd32(SRC(mIdxS)) == zo_rd32(SRC(ip))) { shortCand = 1; match = mIdxS; } }
                else if (tagS) {
                    match = dEntS >> 8; mIdxS = match;
                    if (match > dictStart && sp_rd32(&sp, match) == zo_rd32(SRC(ip))) shortCand = 1;
                }
                if (!shortCand) { ip += ((ip - anchor) >> 8) + 1; continue; }
                /* a short match exists: first try a long match one position later */
                {   const uint32_t hl3 = hash_n(SRC(ip + 1), hl, 8), dTagL3 = hash_n(SRC(ip + 1), dhl, 8);
                    const uint32_t mIdxL3 = hashLong[hl3], dEntL3 = d->hashLong[dTagL3 >> 8];
                    const int tagL3 = (dEntL3 & 255) == (dTagL3 & 255);
                    hashLong[hl3] = curr + 1;
                    if (mIdxL3 >= CE && zo_rd64(SRC(mIdxL3)) == zo_rd64(SRC(ip + 1))) {
                        uint32_t m = mIdxL3;
                        mLength = sp_count(&sp, ip + 9, m + 8) + 8;
                        ip++;
                        offset = ip - m;
                        while (ip > anchor && m > CE && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; }
                        found = 2;
                    } else if (tagL3) {
                        uint32_t m = dEntL3 >> 8;
                        if (m > dictStart && sp_rd64(&sp, m) == zo_rd64(SRC(ip + 1))) {
                            mLength = sp_count(&sp, ip + 1 + 8, m + 8) + 8;
                            ip++;
                            offset = curr + 1 - m;
                            while (ip > anchor && m > dictStart && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; }
                            found = 2;
                        }
                    }
                }
                if (!found) {
                    uint32_t m = match;
                    mLength = sp_count(&sp, ip + 4, m + 4) + 4;
                    offset = curr - mIdxS;
                    if (mIdxS < CE) { while (ip > anchor && m > dictStart && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; } }
                    else { while (ip > anchor && m > CE && sp_byte(&sp, ip - 1) == sp_byte(&sp, m - 1)) { ip--; m--; mLength++; } }
                    found = 2;
                }
            }
            off2 = off1; off1 = offset;
            STORE(ip - anchor, offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = curr + 2;
            hashLong[hash_n(SRC(ins), hl, 8)] = ins;
            hashLong[hash_n(SRC(ip - 2), hl, 8)] = ip - 2;
            hashSmall[hash_n(SRC(ins), hs, mls)] = ins;
            hashSmall[hash_n(SRC(ip - 1), hs, mls)] = ip - 1;
            while (ip <= ilimit) {
                const uint32_t rep2 = ip - off2;
                if (((uint32_t)((CE - 1) - rep2) >= 3) && sp_rd32(&sp, rep2) == zo_rd32(SRC(ip))) {
                    const uint32_t r = sp_count(&sp, ip + 4, rep2 + 4) + 4;
                    uint32_t t = off2; off2 = off1; off1 = t;
                    STORE(0, 1, r);
                    hashSmall[hash_n(SRC(ip), hs, mls)] = ip;
                    hashLong[hash_n(SRC(ip), hl, 8)] = ip;
                    ip += r; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
#undef STORE
    { size_t last = (size_t)(iend - anchor); memcpy(lp, SRC(anchor), last); lp += last; }
#undef SRC
    *litSize = (size_t)(lp - lits);
    return nseq;
}

/* ------------------------------------------------------------------ sequences section */
static unsigned ll_code(uint32_t v) { unsigned c = 35; while (zo_ll_base[c] > v) c--; return c; }
static unsigned ml_code(uint32_t ml) { unsigned c = 52; while (zo_ml_base[c] > ml) c--; return c; }

/* picks basic / rle / compressed exactly like the reference does for strategies below "lazy" on a first block */
static int select_mode(const unsigned* count, unsigned max, unsigned mostFrequent, size_t nbSeq, unsigned defLog, int defaultAllowed, int repeatMode, int strat)
{
    (void)count; (void)max;
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        size_t dynMin = (((size_t)1 << defLog) * (size_t)(10 - strat)) >> 3;     /* mult = 10 - strategy (fast 1, dfast 2) */
        if (repeatMode == 2 && nbSeq < 1000) return 3;         /* set_repeat: the dictionary's table is known to cover everything */
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defLog - 1))) return 0;
    }
    return 2;
}

static size_t build_seq_table(fse_ctab* t, uint8_t* out, int* mode, const uint8_t* codes, size_t nbSeq, unsigned maxCode,
                              unsigned fseLog, const int16_t* defNorm, unsigned defLog, unsigned defMax, int allowDefaultIfMaxLE,
                              const fse_ctab* prevTab, int prevRepeat, int strat)
{
    unsigned count[64] = {0}, max = 0, most = 0;
    for (size_t i = 0; i < nbSeq; i++) count[codes[i]]++;
    for (unsigned s = 0; s <= maxCode; s++) { if (count[s]) max = s; if (count[s] > most) most = count[s]; }
    int defaultAllowed = allowDefaultIfMaxLE < 0 ? 1 : (max <= (unsigned)allowDefaultIfMaxLE);
    *mode = select_mode(count, max, most, nbSeq, defLog, defaultAllowed, prevTab ? prevRepeat : 0, strat);
    if (*mode == 3) { *t = *prevTab; return 0; }
    if (*mode == 1) { fse_build_rle(t, codes[0]); out[0] = codes[0]; return 1; }
    if (*mode == 0) { fse_build_ctab(t, defNorm, defMax, defLog); return 0; }
    unsigned log = fse_optimal_log(fseLog, nbSeq, max, 2);
    size_t n1 = nbSeq;
    if (count[codes[nbSeq - 1]] > 1) { count[codes[nbSeq - 1]]--; n1--; }
    int16_t norm[64];
    fse_normalize(norm, log, count, n1, max, n1 >= 2048);
    size_t h = fse_write_ncount(out, norm, max, log);
    fse_build_ctab(t, norm, max, log);
    return h;
}

/* ------------------------------------------------------------------ one block */
/* returns compressed body size, or 0 when the block must be stored raw */
static size_t zo_compress_block(uint8_t* out, size_t cap, const uint8_t* src, size_t srcSize, const zo_cpar* cp, const zo_cdict* cd)
{
    const zo_entropy* prev = cd ? &cd->ent : NULL;
    if (srcSize < 7) return 0;
    size_t result = 0;
    zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (srcSize / 3 + 8));
    uint8_t* lits = (uint8_t*)malloc(srcSize + 64);
    uint8_t* codes = (uint8_t*)malloc(3 * (srcSize / 3 + 8));
    uint32_t* hashLong = (uint32_t*)malloc(sizeof(uint32_t) << cp->hlog);
    uint32_t* hashSmall = (uint32_t*)malloc(sizeof(uint32_t) << cp->clog);
    fse_ctab* tabs = (fse_ctab*)malloc(3 * sizeof(fse_ctab));
    if (!seqs || !lits || !codes || !hashLong || !hashSmall || !tabs) goto done;
    {
        size_t litSize = 0;
        size_t nbSeq = cd ? zo_dfast_dict(seqs, lits, &litSize, src, srcSize, cp, cd, hashLong, hashSmall)
                          : cp->strat == 1 ? zo_fast(seqs, lits, &litSize, src, srcSize, cp, hashLong)
                          : zo_dfast(seqs, lits, &litSize, src, srcSize, cp, hashLong, hashSmall);
        uint8_t* op = out;
        /* literals stay raw with the fast strategy at a non-zero target length, i.e. negative levels (ZSTD_literalsCompressionIsDisabled, zstd.c:24208) */
        if (cp->strat == 1 && cp->tlen > 0) op += zo_raw_literals(op, lits, litSize, 0, 0);
        else op += zo_compress_literals(op, cap, lits, litSize, nbSeq, prev, NULL, NULL);
        if (nbSeq < 128) *op++ = (uint8_t)nbSeq;
        else if (nbSeq < 0x7F00) { op[0] = (uint8_t)((nbSeq >> 8) + 0x80); op[1] = (uint8_t)nbSeq; op += 2; }
        else { op[0] = 0xFF; zo_wr16(op + 1, (uint16_t)(nbSeq - 0x7F00)); op += 3; }
        if (nbSeq) {
            uint8_t* llc = codes; uint8_t* ofc = codes + nbSeq; uint8_t* mlc = codes + 2 * nbSeq;
            for (size_t i = 0; i < nbSeq; i++) {
                llc[i] = (uint8_t)ll_code(seqs[i].litLength);
                ofc[i] = (uint8_t)zo_highbit(seqs[i].offBase);
                mlc[i] = (uint8_t)ml_code(seqs[i].matchLength);
            }
            uint8_t* seqHead = op++;
            int mLL, mOF, mML; size_t lastCount = 0, h;
            h = build_seq_table(&tabs[0], op, &mLL, llc, nbSeq, 35, 9, zo_ll_defnorm, 6, 35, -1, prev ? &prev->ll : NULL, prev ? prev->llRepeat : 0, cp->strat); if (mLL == 2) lastCount = h; op += h;
            h = build_seq_table(&tabs[1], op, &mOF, ofc, nbSeq, 31, 8, zo_of_defnorm, 5, 28, 28, prev ? &prev->of : NULL, prev ? prev->ofRepeat : 0, cp->strat); if (mOF == 2) lastCount = h; op += h;
            h = build_seq_table(&tabs[2], op, &mML, mlc, nbSeq, 52, 9, zo_ml_defnorm, 6, 52, -1, prev ? &prev->ml : NULL, prev ? prev->mlRepeat : 0, cp->strat); if (mML == 2) lastCount = h; op += h;
            *seqHead = (uint8_t)((mLL << 6) + (mOF << 4) + (mML << 2));
            /* three interleaved states, sequences visited last to first; per sequence OF state, ML state, LL state,
             * then LL, ML, OF extra bits (the decoder reads them in the opposite order) */
            bitw b; bw_init(&b, op, cap - (size_t)(op - out));
            size_t n = nbSeq - 1;
            uint32_t sML = fse_first_state(&tabs[2], mlc[n]), sOF = fse_first_state(&tabs[1], ofc[n]), sLL = fse_first_state(&tabs[0], llc[n]);
            bw_add(&b, seqs[n].litLength, zo_ll_bits[llc[n]]);
            bw_add(&b, seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]);
            bw_add(&b, seqs[n].offBase, ofc[n]);
            while (n-- > 0) {
                sOF = fse_encode(&tabs[1], &b, sOF, ofc[n]);
                sML = fse_encode(&tabs[2], &b, sML, mlc[n]);
                sLL = fse_encode(&tabs[0], &b, sLL, llc[n]);
                bw_add(&b, seqs[n].litLength, zo_ll_bits[llc[n]]);
                bw_add(&b, seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]);
                bw_add(&b, seqs[n].offBase, ofc[n]);
            }
            bw_add(&b, sML, tabs[2].log); bw_add(&b, sOF, tabs[1].log); bw_add(&b, sLL, tabs[0].log);
            size_t bs = bw_close(&b);
            if (bs == 0) goto done;
            op += bs;
            if (lastCount && lastCount + bs < 4) goto done;       /* old-decoder workaround: store raw */
        }
        size_t cSize = (size_t)(op - out);
        if (cSize >= srcSize - ((srcSize >> 6) + 2)) goto done;    /* not enough gain */
        result = cSize;
    }
done:
    free(seqs); free(lits); free(codes); free(hashLong); free(hashSmall); free(tabs);
    return result;
}


/* ------------------------------------------------------------------ multi-block frames (sources above 128 KiB)
 * Restates ZSTD_compress_frameChunk (zstd.c:27545), ZSTD_optimalBlockSize (:27506) with the pre-splitter (:22263-22502),
 * ZSTD_compressBlock_internal (:27337) and the state that survives a block: hash tables, two repcodes and the literals
 * Huffman table -- each of them advanced ONLY when the block was emitted compressed (:27391-27393). Without a dictionary and
 * below the lazy strategies the FSE tables never repeat (a table left by a block is "check", which this strategy class
 * ignores, :21252-21331), so they are not carried. When the source outgrows the window, the lowest usable index follows the end
 * of each block (ZSTD_window_enforceMaxDist / ZSTD_getLowestPrefixIndex). */
static uint64_t fp_distance(const unsigned* a, uint64_t na, const unsigned* b, uint64_t nb, int bins)
{
    uint64_t d = 0;
    for (int i = 0; i < bins; i++) { int64_t x = (int64_t)a[i] * (int64_t)nb - (int64_t)b[i] * (int64_t)na; d += (uint64_t)(x < 0 ? -x : x); }
    return d;
}
static int fp_too_different(const unsigned* ref, uint64_t nref, const unsigned* nw, uint64_t nnew, int penalty, int bins)
{
    uint64_t p50 = nref * nnew, dev = fp_distance(ref, nref, nw, nnew, bins);
    return dev >= p50 * (uint64_t)(14 + penalty) / 16;
}
/* where to end a full 128 KiB block once the frame has shown savings */
static size_t zo_split_block(const uint8_t* p, int strat)
{
    const size_t B = 128 << 10;
    if (strat == 1) {           /* fast: compare byte histograms of the first, last (and middle) 512 bytes */
        unsigned first[256], last[256], mid[256];
        memset(first, 0, sizeof first); memset(last, 0, sizeof last); memset(mid, 0, sizeof mid);
        for (int i = 0; i < 512; i++) { first[p[i]]++; last[p[B - 512 + i]]++; }
        if (!fp_too_different(first, 512, last, 512, 0, 256)) return B;
        for (int i = 0; i < 512; i++) mid[p[B / 2 - 256 + i]]++;
        uint64_t db = fp_distance(first, 512, mid, 512, 256), de = fp_distance(last, 512, mid, 512, 256);
        int64_t diff = (int64_t)db - (int64_t)de; if (diff < 0) diff = -diff;
        if ((uint64_t)diff < 512ull * 512 / 3) return 64 << 10;
        return db > de ? (32 << 10) : (96 << 10);
    }
    /* double-fast: 8 KiB chunks, byte histogram of every 43rd position, split at the first chunk that differs from the past */
    unsigned past[256], cur[256]; uint64_t npast = 0;
    memset(past, 0, sizeof past);
    const size_t C = 8 << 10, limit = C - 2 + 1;          /* positions 0 .. C-2 step 43 */
    for (size_t n = 0; n < limit; n += 43) past[p[n]]++;
    npast = limit / 43;
    int penalty = 3;
    for (size_t pos = C; pos <= B - C; pos += C) {
        memset(cur, 0, sizeof cur);
        for (size_t n = 0; n < limit; n += 43) cur[p[pos + n]]++;
        const uint64_t ncur = limit / 43;
        if (fp_too_different(past, npast, cur, ncur, penalty, 256)) return pos;
        for (int i = 0; i < 256; i++) past[i] += cur[i];
        npast += ncur;
        if (penalty > 0) penalty--;
    }
    return B;
}

typedef struct { uint32_t rep[2]; huf_ctab huf; int hufRepeat; } zo_blockstate;

/* one block of a multi-block frame; returns the body size (0: store raw, 1: RLE block, body[0] = the byte) */
static size_t zo_compress_block_g(uint8_t* out, size_t cap, const uint8_t* frame, const uint8_t* src, size_t srcSize, const zo_cpar* cp,
                                  uint32_t* hashLong, uint32_t* hashSmall, zo_blockstate* st, int firstBlock,
                                  zo_seq* seqs, uint8_t* lits, uint8_t* codes, fse_ctab* tabs)
{
    if (srcSize < 7) return 0;                       /* too small to try: the search is skipped as well */
    zo_blockstate next = *st;
    size_t litSize = 0, cSize = 0;
    const size_t nbSeq = cp->strat == 1 ? zo_fast_g(seqs, lits, &litSize, frame, src, srcSize, cp, hashLong, next.rep)
                                        : zo_dfast_g(seqs, lits, &litSize, frame, src, srcSize, cp, hashLong, hashSmall, next.rep);
    {
        uint8_t* op = out;
        zo_entropy prev; memset(&prev, 0, sizeof prev); prev.huf = st->huf; prev.hufRepeat = st->hufRepeat;
        int usedNew = 0; huf_ctab nt;
        if (cp->strat == 1 && cp->tlen > 0) op += zo_raw_literals(op, lits, litSize, 0, 0);
        else op += zo_compress_literals(op, cap, lits, litSize, nbSeq, &prev, &nt, &usedNew);
        if (usedNew) { next.huf = nt; next.hufRepeat = 1; }
        if (nbSeq < 128) *op++ = (uint8_t)nbSeq;
        else if (nbSeq < 0x7F00) { op[0] = (uint8_t)((nbSeq >> 8) + 0x80); op[1] = (uint8_t)nbSeq; op += 2; }
        else { op[0] = 0xFF; zo_wr16(op + 1, (uint16_t)(nbSeq - 0x7F00)); op += 3; }
        int ok = 1;
        if (nbSeq) {
            uint8_t* llc = codes; uint8_t* ofc = codes + nbSeq; uint8_t* mlc = codes + 2 * nbSeq;
            for (size_t i = 0; i < nbSeq; i++) { llc[i] = (uint8_t)ll_code(seqs[i].litLength); ofc[i] = (uint8_t)zo_highbit(seqs[i].offBase); mlc[i] = (uint8_t)ml_code(seqs[i].matchLength); }
            uint8_t* seqHead = op++;
            int mLL, mOF, mML; size_t lastCount = 0, h;
            h = build_seq_table(&tabs[0], op, &mLL, llc, nbSeq, 35, 9, zo_ll_defnorm, 6, 35, -1, NULL, 0, cp->strat); if (mLL == 2) lastCount = h; op += h;
            h = build_seq_table(&tabs[1], op, &mOF, ofc, nbSeq, 31, 8, zo_of_defnorm, 5, 28, 28, NULL, 0, cp->strat); if (mOF == 2) lastCount = h; op += h;
            h = build_seq_table(&tabs[2], op, &mML, mlc, nbSeq, 52, 9, zo_ml_defnorm, 6, 52, -1, NULL, 0, cp->strat); if (mML == 2) lastCount = h; op += h;
            *seqHead = (uint8_t)((mLL << 6) + (mOF << 4) + (mML << 2));
            bitw b; bw_init(&b, op, cap - (size_t)(op - out));
            size_t n = nbSeq - 1;
            uint32_t sML = fse_first_state(&tabs[2], mlc[n]), sOF = fse_first_state(&tabs[1], ofc[n]), sLL = fse_first_state(&tabs[0], llc[n]);
            bw_add(&b, seqs[n].litLength, zo_ll_bits[llc[n]]); bw_add(&b, seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]); bw_add(&b, seqs[n].offBase, ofc[n]);
            while (n-- > 0) {
                sOF = fse_encode(&tabs[1], &b, sOF, ofc[n]); sML = fse_encode(&tabs[2], &b, sML, mlc[n]); sLL = fse_encode(&tabs[0], &b, sLL, llc[n]);
                bw_add(&b, seqs[n].litLength, zo_ll_bits[llc[n]]); bw_add(&b, seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]); bw_add(&b, seqs[n].offBase, ofc[n]);
            }
            bw_add(&b, sML, tabs[2].log); bw_add(&b, sOF, tabs[1].log); bw_add(&b, sLL, tabs[0].log);
            size_t bs = bw_close(&b);
            if (bs == 0) ok = 0;
            op += bs;
            if (ok && lastCount && lastCount + bs < 4) ok = 0;
        }
        if (ok) { cSize = (size_t)(op - out); if (cSize >= srcSize - ((srcSize >> 6) + 2)) cSize = 0; }
    }
    if (!firstBlock && cSize < 25) {                 /* a later block made of one byte value becomes an RLE block (zstd.c:27373-27384) */
        int same = 1; for (size_t i = 1; i < srcSize; i++) if (src[i] != src[0]) { same = 0; break; }
        if (same) { out[0] = src[0]; cSize = 1; }
    }
    if (cSize > 1) *st = next;                       /* repcodes and the Huffman table advance only with a compressed block */
    return cSize;
}

static int64_t zo_compress_frame_multi(uint8_t* dst, size_t dstCap, const uint8_t* src, size_t srcSize, const zo_cpar* cp, unsigned flags)
{
    (void)dstCap;
    if (srcSize >= (1u << 31)) return -ZO_E_PARAM_UNSUPPORTED;                   /* index overflow correction is not restated */
    const int contentSize = (flags & ZO_F_CONTENTSIZE) != 0, checksum = (flags & ZO_F_CHECKSUM) != 0;
    size_t pos = 0;
    zo_wr32(dst, ZO_MAGIC); pos = 4;
    const uint32_t windowSize = 1u << cp->wlog;
    const int single = contentSize && windowSize >= srcSize;
    const unsigned fcsCode = contentSize ? (srcSize >= 256) + (srcSize >= 65536 + 256) + (srcSize >= 0xFFFFFFFFu) : 0;
    dst[pos++] = (uint8_t)((checksum << 2) + (single << 5) + (fcsCode << 6));
    if (!single) dst[pos++] = (uint8_t)((cp->wlog - 10) << 3);
    if (fcsCode == 0) { if (single) dst[pos++] = (uint8_t)srcSize; }
    else if (fcsCode == 1) { zo_wr16(dst + pos, (uint16_t)(srcSize - 256)); pos += 2; }
    else if (fcsCode == 2) { zo_wr32(dst + pos, (uint32_t)srcSize); pos += 4; }
    else { zo_wr64(dst + pos, (uint64_t)srcSize); pos += 8; }
    uint32_t* hashLong = (uint32_t*)calloc((size_t)1 << cp->hlog, 4);
    uint32_t* hashSmall = (uint32_t*)calloc((size_t)1 << cp->clog, 4);
    zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (ZO_BLOCK_MAX / 3 + 8));
    uint8_t* lits = (uint8_t*)malloc(ZO_BLOCK_MAX + 64);
    uint8_t* codes = (uint8_t*)malloc(3 * (ZO_BLOCK_MAX / 3 + 8));
    fse_ctab* tabs = (fse_ctab*)malloc(3 * sizeof(fse_ctab));
    int64_t result = -ZO_E_MEMORY;
    if (hashLong && hashSmall && seqs && lits && codes && tabs) {
        zo_blockstate st; memset(&st, 0, sizeof st); st.rep[0] = 1; st.rep[1] = 4;
        size_t ip = 0; int64_t savings = 0; int first = 1;
        while (ip < srcSize) {
            const size_t remaining = srcSize - ip;
            size_t blockSize = remaining < ZO_BLOCK_MAX ? remaining : ZO_BLOCK_MAX;
            if (remaining >= ZO_BLOCK_MAX && savings >= 3) blockSize = zo_split_block(src + ip, cp->strat);
            const int last = blockSize == remaining;
            const size_t c = zo_compress_block_g(dst + pos + 3, zo_compress_bound(blockSize) + 64, src, src + ip, blockSize, cp, hashLong, hashSmall, &st, first,
                                                 seqs, lits, codes, tabs);
            size_t total;
            if (c == 0) { zo_wr24(dst + pos, (uint32_t)(last + (0u << 1) + (blockSize << 3))); memcpy(dst + pos + 3, src + ip, blockSize); total = 3 + blockSize; }
            else if (c == 1) { zo_wr24(dst + pos, (uint32_t)(last + (1u << 1) + (blockSize << 3))); total = 4; }
            else { zo_wr24(dst + pos, (uint32_t)(last + (2u << 1) + (c << 3))); total = 3 + c; }
            savings += (int64_t)blockSize - (int64_t)total;
            pos += total; ip += blockSize; first = 0;
        }
        if (checksum) { zo_wr32(dst + pos, (uint32_t)zo_xxh64(src, srcSize, 0)); pos += 4; }
        result = (int64_t)pos;
    }
    free(hashLong); free(hashSmall); free(seqs); free(lits); free(codes); free(tabs);
    return result;
}

/* ------------------------------------------------------------------ frame */
size_t zo_compress_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }

int64_t zo_compress_frame(void* dstv, size_t dstCap, const void* srcv, size_t srcSize, int level, unsigned flags,
                          const void* dict, size_t dictSize)
{
    uint8_t* dst = (uint8_t*)dstv; const uint8_t* src = (const uint8_t*)srcv;
    if (dstCap < zo_compress_bound(srcSize)) return -ZO_E_DST_TOO_SMALL;
    zo_cpar cp; int e = zo_get_cparams(&cp, level, srcSize); if (e < 0) return e;
    if (srcSize > ZO_BLOCK_MAX) {
        if ((dict && dictSize) || (cp.strat != 1 && cp.strat != 2)) return -ZO_E_PARAM_UNSUPPORTED;
        return zo_compress_frame_multi(dst, dstCap, src, srcSize, &cp, flags);
    }
    zo_cdict* cd = NULL;
    if (dict && dictSize) {
        /* only the attached-CDict mode (sources <= 16 KiB for double-fast, zstd.c:25235-25276) is restated */
        if (dictSize < 8 || srcSize > 16 * 1024) return -ZO_E_PARAM_UNSUPPORTED;
        cd = zo_cdict_create((const uint8_t*)dict, dictSize, level, &e);
        if (!cd) return e;
        /* working tables: the CDict's parameters shrunk to the source; the frame keeps the window log chosen for the source */
        zo_cpar w = cd->cp;
        uint32_t t = (uint32_t)srcSize; int srcLog = (t < 64) ? 6 : zo_highbit(t - 1) + 1;
        if (w.wlog > srcLog) w.wlog = srcLog;
        if (w.hlog > w.wlog + 1) w.hlog = w.wlog + 1;
        if (w.clog > w.wlog) w.clog = w.wlog;
        w.wlog = cp.wlog;
        cp = w;
    }
    if (cp.strat != 2 && !(cp.strat == 1 && !cd)) { zo_cdict_free(cd); return -ZO_E_PARAM_UNSUPPORTED; }
    /* frame header */
    size_t pos = 0;
    const int contentSize = (flags & ZO_F_CONTENTSIZE) != 0, checksum = (flags & ZO_F_CHECKSUM) != 0;
    const uint32_t windowSize = 1u << cp.wlog;
    const int single = contentSize && windowSize >= srcSize;
    const unsigned fcsCode = contentSize ? (srcSize >= 256) + (srcSize >= 65536 + 256) + (srcSize >= 0xFFFFFFFFu) : 0;
    zo_wr32(dst, ZO_MAGIC); pos = 4;
    const uint32_t dictID = (cd && (flags & ZO_F_DICTID)) ? cd->dictID : 0;
    const unsigned dictCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    dst[pos++] = (uint8_t)(dictCode + (checksum << 2) + (single << 5) + (fcsCode << 6));
    if (!single) dst[pos++] = (uint8_t)((cp.wlog - 10) << 3);
    if (dictCode == 1) dst[pos++] = (uint8_t)dictID;
    else if (dictCode == 2) { zo_wr16(dst + pos, (uint16_t)dictID); pos += 2; }
    else if (dictCode == 3) { zo_wr32(dst + pos, dictID); pos += 4; }
    if (fcsCode == 0) { if (single) dst[pos++] = (uint8_t)srcSize; }
    else if (fcsCode == 1) { zo_wr16(dst + pos, (uint16_t)(srcSize - 256)); pos += 2; }
    else if (fcsCode == 2) { zo_wr32(dst + pos, (uint32_t)srcSize); pos += 4; }
    else { zo_wr64(dst + pos, srcSize); pos += 8; }
    /* the single block (or the empty last block of an empty frame) */
    if (srcSize == 0) { zo_wr24(dst + pos, 1); pos += 3; }
    else {
        size_t c = zo_compress_block(dst + pos + 3, dstCap - pos - 3, src, srcSize, &cp, cd);
        if (c == 0) { zo_wr24(dst + pos, 1 + (0 << 1) + ((uint32_t)srcSize << 3)); memcpy(dst + pos + 3, src, srcSize); pos += 3 + srcSize; }
        else { zo_wr24(dst + pos, 1 + (2 << 1) + ((uint32_t)c << 3)); pos += 3 + c; }
    }
    if (checksum) { zo_wr32(dst + pos, (uint32_t)zo_xxh64(src, srcSize, 0)); pos += 4; }
    zo_cdict_free(cd);
    return (int64_t)pos;
}

/* ------------------------------------------------------------------ test hooks (tests/test_emu_kernels.py): the table builders on their own,
 * so that the wave-parallel builders of the HIP encoder can be checked against this serial restatement on arbitrary histograms */
int zo_test_fse_tables(const unsigned* count, unsigned maxSym, unsigned total, unsigned log, int useLowProb,
                       int16_t* normOut /* 64 */, uint8_t* ncountOut /* 512 */, unsigned* ncountSize, uint16_t* cellOf /* 66 */, uint16_t* next /* 512 */)
{
    int16_t norm[64]; memset(norm, 0, sizeof norm);
    if (fse_normalize(norm, log, count, total, maxSym, useLowProb) < 0) return -1;
    memcpy(normOut, norm, sizeof norm);
    *ncountSize = (unsigned)fse_write_ncount(ncountOut, norm, maxSym, log);
    fse_ctab t; memset(&t, 0, sizeof t);
    fse_build_ctab(&t, norm, maxSym, log);
    for (unsigned s = 0; s <= maxSym + 1; s++) cellOf[s] = t.cellOf[s];
    for (unsigned u = 0; u < (1u << log); u++) next[u] = t.next[u];
    return 0;
}
unsigned zo_test_huf_build(const unsigned* count, unsigned maxSym, unsigned maxBits, uint8_t* nbBits /* 256 */, uint16_t* code /* 256 */)
{
    huf_ctab ct; memset(&ct, 0, sizeof ct);
    const unsigned log = huf_build(&ct, count, maxSym, maxBits);
    memcpy(nbBits, ct.nbBits, 256); memcpy(code, ct.code, 512);
    return log;
}
