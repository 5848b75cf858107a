/* oracle/zo_decode.c -- CPU oracle: zstd frame DECODER (TEST INFRASTRUCTURE ONLY, see zo_common.h).
 *
 * Restates the decode side of the reference hot path:
 *   frame header      <- zstd.c:43668 ZSTD_getFrameHeader_advanced, :43923 ZSTD_decodeFrameHeader
 *   block loop        <- zstd.c:44174 ZSTD_decompressFrame, :45696 ZSTD_getcBlockSize
 *   literals          <- zstd.c:45767 ZSTD_decodeLiteralsBlock, :39651 HUF_readDTableX1_wksp, :3448 HUF_readStats
 *   sequence tables   <- zstd.c:46328 ZSTD_decodeSeqHeaders, :3256 FSE_readNCount_body, :46118 ZSTD_buildFSETable_body
 *   sequences         <- zstd.c:47248 ZSTD_decompressSequences_body, :46862 ZSTD_decodeSequence, :46634 ZSTD_execSequence
 *   dictionary        <- zstd.c:44673 ZSTD_loadDEntropy
 * written from the format (RFC 8878); any correct decoder yields identical bytes, so this file is
 * deliberately the simplest possible scalar formulation (bit positions as integers, byte copies).
 */
#include "zstd_oracle.h"
#include "zo_common.h"
#include <stdlib.h>

const uint32_t zo_ll_base[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,
                                 1024,2048,4096,8192,16384,32768,65536};
const uint8_t zo_ll_bits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
const uint32_t zo_ml_base[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,
                                 33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
const uint8_t zo_ml_bits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
const int16_t zo_ll_defnorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
const int16_t zo_ml_defnorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                   1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
const int16_t zo_of_defnorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

/* ------------------------------------------------------------------ XXH64 */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; }
static inline uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround(0, v)) * P1 + P4; }
uint64_t zo_xxh64(const void* data, size_t len, uint64_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xround(v1, zo_rd64(p)); v2 = xround(v2, zo_rd64(p + 8));
            v3 = xround(v3, zo_rd64(p + 16)); v4 = xround(v4, zo_rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xround(0, zo_rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)zo_rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p++) * P5; h = rotl64(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ bit readers */
/* forward LSB-first reader (FSE table descriptions) */
typedef struct { const uint8_t* p; size_t size; size_t bitpos; } fwd_bits;
static uint32_t fwd_peek(const fwd_bits* b, int n)
{
    uint64_t v = 0; size_t byte = b->bitpos >> 3;
    for (int i = 0; i < 8; i++) if (byte + i < b->size) v |= (uint64_t)b->p[byte + i] << (8 * i);
    return (uint32_t)((v >> (b->bitpos & 7)) & ((1ull << n) - 1));
}

/* backward reader: the stream is read from its last byte; `bits` = number of not-yet-read bits.
 * reading below bit 0 yields zeros and drives `bits` negative (== libzstd's "overflow" state). */
typedef struct { const uint8_t* p; int64_t bits; } bwd_bits;
static int bwd_init(bwd_bits* b, const uint8_t* p, size_t size)
{
    if (size == 0) return -ZO_E_SRC_SIZE_WRONG;
    if (p[size - 1] == 0) return -ZO_E_CORRUPTION; /* end mark missing */
    b->p = p;
    b->bits = (int64_t)size * 8 - (8 - zo_highbit(p[size - 1]));
    return 0;
}
static uint64_t bwd_peek_at(const bwd_bits* b, int64_t pos, int n) /* bits [pos, pos+n), n<=32 */
{
    uint64_t v = 0;
    if (n == 0) return 0;
    for (int i = 0; i < n; i++) {
        int64_t bp = pos + i;
        if (bp >= 0) v |= (uint64_t)((b->p[bp >> 3] >> (bp & 7)) & 1) << i;
    }
    return v;
}
static inline uint64_t bwd_read(bwd_bits* b, int n) { b->bits -= n; return bwd_peek_at(b, b->bits, n); }
static inline uint64_t bwd_peek(const bwd_bits* b, int n) { return bwd_peek_at(b, b->bits - n, n); }

/* ------------------------------------------------------------------ FSE */
int zo_fse_read_ncount(int16_t* norm, unsigned* maxSymbol, unsigned* tableLog, const uint8_t* src, size_t srcSize)
{
    fwd_bits b = { src, srcSize, 0 };
    if (srcSize < 1) return -ZO_E_SRC_SIZE_WRONG;
    int al = (int)fwd_peek(&b, 4) + 5; b.bitpos += 4;
    if (al > 15) return -ZO_E_TABLELOG_TOO_LARGE;
    *tableLog = (unsigned)al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbBits = al + 1;
    unsigned sym = 0, maxS = *maxSymbol;
    int prev0 = 0;
    while (remaining > 1 && sym <= maxS) {
        if (prev0) {
            for (;;) {
                unsigned r = fwd_peek(&b, 2); b.bitpos += 2;
                for (unsigned k = 0; k < r && sym <= maxS; k++) norm[sym++] = 0;
                if (r != 3) break;
                if (b.bitpos > srcSize * 8) return -ZO_E_CORRUPTION;
            }
            if (sym > maxS) return -ZO_E_MAXSYMBOL_TOO_SMALL;
        }
        int max = (2 * threshold - 1) - remaining;
        int count;
        int low = (int)fwd_peek(&b, nbBits - 1);
        if (low < max) { count = low; b.bitpos += nbBits - 1; }
        else {
            count = (int)fwd_peek(&b, nbBits);
            if (count >= threshold) count -= max;
            b.bitpos += nbBits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        if (remaining < 1) return -ZO_E_CORRUPTION;
        norm[sym++] = (int16_t)count;
        prev0 = (count == 0);
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if (b.bitpos > srcSize * 8 + 0) { /* tolerate reading padding of the final byte only */
            if (((b.bitpos + 7) >> 3) > srcSize) return -ZO_E_CORRUPTION;
        }
    }
    if (remaining != 1) return -ZO_E_CORRUPTION;
    if (sym > maxS + 1) return -ZO_E_MAXSYMBOL_TOO_SMALL;
    for (unsigned s = sym; s <= maxS; s++) norm[s] = 0;
    *maxSymbol = sym - 1;
    size_t used = (b.bitpos + 7) >> 3;
    if (used > srcSize) return -ZO_E_SRC_SIZE_WRONG;
    return (int)used;
}

typedef struct { uint8_t sym; uint8_t nbBits; uint16_t base; } fse_cell;
typedef struct { fse_cell cell[512]; int log; } fse_dtable;

static int fse_build_dtable(fse_dtable* dt, const int16_t* norm, unsigned maxSymbol, unsigned tableLog)
{
    uint16_t next[256];
    unsigned size = 1u << tableLog, high = size - 1;
    if (tableLog > 9) return -ZO_E_TABLELOG_TOO_LARGE;
    dt->log = (int)tableLog;
    for (unsigned s = 0; s <= maxSymbol; s++) {
        if (norm[s] == -1) { dt->cell[high--].sym = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    unsigned step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (unsigned s = 0; s <= maxSymbol; s++) {
        for (int i = 0; i < norm[s]; i++) {
            dt->cell[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return -ZO_E_CORRUPTION;
    for (unsigned u = 0; u < size; u++) {
        unsigned s = dt->cell[u].sym, x = next[s]++;
        int nb = (int)tableLog - zo_highbit(x);
        dt->cell[u].nbBits = (uint8_t)nb;
        dt->cell[u].base = (uint16_t)((x << nb) - size);
    }
    return 0;
}
static void fse_build_rle(fse_dtable* dt, uint8_t sym)
{
    dt->log = 0; dt->cell[0].sym = sym; dt->cell[0].nbBits = 0; dt->cell[0].base = 0;
}

/* ------------------------------------------------------------------ Huffman */
typedef struct { uint8_t sym, nb; } huf_cell;
typedef struct { huf_cell cell[4096]; int log; } huf_dtable;

/* weights -> decoding table. nWeights explicit weights; the last one is implied. */
static int huf_build_dtable(huf_dtable* ht, uint8_t* w, unsigned nWeights)
{
    uint32_t total = 0;
    if (nWeights == 0 || nWeights > 255) return -ZO_E_CORRUPTION;
    for (unsigned i = 0; i < nWeights; i++) {
        if (w[i] > 12) return -ZO_E_CORRUPTION;
        total += w[i] ? (1u << (w[i] - 1)) : 0;
    }
    if (total == 0) return -ZO_E_CORRUPTION;
    int log = zo_highbit(total) + 1;
    if (log > 12) return -ZO_E_CORRUPTION;
    uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1)) return -ZO_E_CORRUPTION; /* must be a clean power of two */
    w[nWeights] = (uint8_t)(zo_highbit(rest) + 1);
    unsigned n = nWeights + 1;
    /* libzstd requires at least two symbols of weight 1 parity: rankStats[1] >= 2 and even */
    unsigned w1 = 0; for (unsigned i = 0; i < n; i++) w1 += (w[i] == 1);
    if (w1 < 2 || (w1 & 1)) return -ZO_E_CORRUPTION;
    ht->log = log;
    uint32_t pos = 0;
    for (int wt = 1; wt <= log; wt++) {
        for (unsigned s = 0; s < n; s++) if (w[s] == wt) {
            uint32_t len = 1u << (wt - 1);
            for (uint32_t k = 0; k < len; k++) { ht->cell[pos + k].sym = (uint8_t)s; ht->cell[pos + k].nb = (uint8_t)(log + 1 - wt); }
            pos += len;
        }
    }
    return 0;
}

/* reads a Huffman tree description; returns bytes consumed */
static int huf_read_table(huf_dtable* ht, const uint8_t* src, size_t srcSize)
{
    uint8_t w[256];
    unsigned n = 0;
    if (srcSize < 1) return -ZO_E_SRC_SIZE_WRONG;
    unsigned hb = src[0];
    size_t used;
    if (hb >= 128) { /* direct 4-bit weights */
        n = hb - 127;
        used = 1 + (n + 1) / 2;
        if (used > srcSize) return -ZO_E_SRC_SIZE_WRONG;
        for (unsigned i = 0; i < n; i++) w[i] = (i & 1) ? (src[1 + i / 2] & 15) : (src[1 + i / 2] >> 4);
    } else {     /* FSE-compressed weights, two interleaved states */
        used = 1 + hb;
        if (used > srcSize) return -ZO_E_SRC_SIZE_WRONG;
        int16_t norm[256]; unsigned maxS = 255, tl;
        int hdr = zo_fse_read_ncount(norm, &maxS, &tl, src + 1, hb);
        if (hdr < 0) return hdr;
        if (tl > 6) return -ZO_E_TABLELOG_TOO_LARGE;
        fse_dtable dt;
        int e = fse_build_dtable(&dt, norm, maxS, tl); if (e < 0) return e;
        bwd_bits b; e = bwd_init(&b, src + 1 + hdr, hb - (size_t)hdr); if (e < 0) return e;
        unsigned s1 = (unsigned)bwd_read(&b, (int)tl), s2 = (unsigned)bwd_read(&b, (int)tl);
        if (b.bits < 0) return -ZO_E_CORRUPTION;
        for (;;) {
            if (n > 253) return -ZO_E_CORRUPTION;
            w[n++] = dt.cell[s1].sym; s1 = dt.cell[s1].base + (unsigned)bwd_read(&b, dt.cell[s1].nbBits);
            if (b.bits < 0) { w[n++] = dt.cell[s2].sym; break; }
            if (n > 253) return -ZO_E_CORRUPTION;
            w[n++] = dt.cell[s2].sym; s2 = dt.cell[s2].base + (unsigned)bwd_read(&b, dt.cell[s2].nbBits);
            if (b.bits < 0) { w[n++] = dt.cell[s1].sym; break; }
        }
    }
    int e = huf_build_dtable(ht, w, n);
    if (e < 0) return e;
    return (int)used;
}

/* tree description -> complete weight list (implied last weight included); shared with the encoder's dictionary loader.
 * Returns bytes consumed; *count = number of symbols, *log = table log. */
int zo_huf_read_weights(uint8_t* w, unsigned* count, unsigned* log, const uint8_t* src, size_t srcSize)
{
    huf_dtable* ht = (huf_dtable*)malloc(sizeof(huf_dtable));
    if (!ht) return -ZO_E_MEMORY;
    int r = huf_read_table(ht, src, srcSize);
    if (r >= 0) {
        /* recover weights from the decode table: nb = log + 1 - weight for every symbol present */
        unsigned maxSym = 0;
        memset(w, 0, 256);
        for (unsigned i = 0; i < (1u << ht->log); i++) { unsigned s = ht->cell[i].sym; w[s] = (uint8_t)(ht->log + 1 - ht->cell[i].nb); if (s > maxSym) maxSym = s; }
        *count = maxSym + 1; *log = (unsigned)ht->log;
    }
    free(ht);
    return r;
}

static int huf_decode_stream(uint8_t* dst, size_t n, const uint8_t* src, size_t srcSize, const huf_dtable* ht)
{
    bwd_bits b; int e = bwd_init(&b, src, srcSize); if (e < 0) return -ZO_E_CORRUPTION;
    for (size_t i = 0; i < n; i++) {
        huf_cell c = ht->cell[bwd_peek(&b, ht->log)];
        dst[i] = c.sym; b.bits -= c.nb;
    }
    return b.bits == 0 ? 0 : -ZO_E_CORRUPTION;
}

/* ------------------------------------------------------------------ decoder state */
typedef struct {
    huf_dtable huf; int hufValid;
    fse_dtable ll, of, ml; int llValid, ofValid, mlValid;
    uint32_t rep[3];
    const uint8_t* dict; size_t dictSize;   /* content part only */
    uint32_t dictID;
    uint8_t* lit;                            /* 128 KiB + slack literal buffer */
} zo_dctx;

static int load_dict(zo_dctx* d, const uint8_t* dict, size_t dictSize)
{
    d->dict = dict; d->dictSize = dictSize; d->dictID = 0;
    if (dictSize < 8 || zo_rd32(dict) != ZO_DICT_MAGIC) return 0; /* raw content dictionary */
    d->dictID = zo_rd32(dict + 4);
    const uint8_t* p = dict + 8; const uint8_t* end = dict + dictSize;
    int r = huf_read_table(&d->huf, p, (size_t)(end - p)); if (r < 0) return -ZO_E_DICT_CORRUPTED;
    p += r; d->hufValid = 1;
    int16_t norm[64]; unsigned maxS, tl;
    maxS = ZO_MAXOFF; r = zo_fse_read_ncount(norm, &maxS, &tl, p, (size_t)(end - p));
    if (r < 0 || tl > ZO_OF_LOGMAX || fse_build_dtable(&d->of, norm, maxS, tl) < 0) return -ZO_E_DICT_CORRUPTED;
    p += r; d->ofValid = 1;
    maxS = ZO_MAXML; r = zo_fse_read_ncount(norm, &maxS, &tl, p, (size_t)(end - p));
    if (r < 0 || tl > ZO_ML_LOGMAX || fse_build_dtable(&d->ml, norm, maxS, tl) < 0) return -ZO_E_DICT_CORRUPTED;
    p += r; d->mlValid = 1;
    maxS = ZO_MAXLL; r = zo_fse_read_ncount(norm, &maxS, &tl, p, (size_t)(end - p));
    if (r < 0 || tl > ZO_LL_LOGMAX || fse_build_dtable(&d->ll, norm, maxS, tl) < 0) return -ZO_E_DICT_CORRUPTED;
    p += r; d->llValid = 1;
    if (p + 12 > end) return -ZO_E_DICT_CORRUPTED;
    size_t content = (size_t)(end - (p + 12));
    for (int i = 0; i < 3; i++) {
        uint32_t rep = zo_rd32(p + 4 * i);
        if (rep == 0 || rep > content) return -ZO_E_DICT_CORRUPTED;
        d->rep[i] = rep;
    }
    p += 12;
    d->dict = p; d->dictSize = content;
    return 0;
}

/* ------------------------------------------------------------------ literals section */
static int decode_literals(zo_dctx* d, const uint8_t* src, size_t srcSize, size_t* litSize, size_t blockMax)
{
    if (srcSize < 1) return -ZO_E_CORRUPTION;
    unsigned type = src[0] & 3, fmt = (src[0] >> 2) & 3;
    size_t hdr, regen, csize = 0;
    int four = 0;
    if (type < 2) { /* raw / rle */
        switch (fmt) {
        case 0: case 2: hdr = 1; regen = src[0] >> 3; break;
        case 1: hdr = 2; if (srcSize < 2) return -ZO_E_CORRUPTION; regen = zo_rd16(src) >> 4; break;
        default: hdr = 3; if (srcSize < 3) return -ZO_E_CORRUPTION; regen = zo_rd24(src) >> 4; break;
        }
        if (regen > blockMax) return -ZO_E_CORRUPTION;
        if (type == 0) {
            if (hdr + regen > srcSize) return -ZO_E_CORRUPTION;
            memcpy(d->lit, src + hdr, regen);
            *litSize = regen; return (int)(hdr + regen);
        }
        if (hdr + 1 > srcSize) return -ZO_E_CORRUPTION;
        memset(d->lit, src[hdr], regen);
        *litSize = regen; return (int)(hdr + 1);
    }
    if (srcSize < 5) return -ZO_E_CORRUPTION; /* libzstd demands 5 readable bytes here (zstd.c:45780) */
    switch (fmt) {
    case 0: case 1: { uint32_t v = zo_rd32(src); hdr = 3; four = fmt; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; break; }
    case 2: { uint32_t v = zo_rd32(src); hdr = 4; four = 1; regen = (v >> 4) & 0x3FFF; csize = v >> 18; break; }
    default: { uint32_t v = zo_rd32(src); hdr = 5; four = 1; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) + ((size_t)src[4] << 10); break; }
    }
    if (regen > blockMax) return -ZO_E_CORRUPTION;
    if (!four && regen == 0) return -ZO_E_CORRUPTION;
    if (four && regen < 6) return -ZO_E_CORRUPTION;
    if (hdr + csize > srcSize) return -ZO_E_CORRUPTION;
    const uint8_t* p = src + hdr; size_t left = csize;
    if (type == 3) { if (!d->hufValid) return -ZO_E_DICT_CORRUPTED; }
    else {
        int r = huf_read_table(&d->huf, p, left); if (r < 0) return -ZO_E_CORRUPTION;
        p += r; left -= (size_t)r; d->hufValid = 1;
    }
    if (!four) {
        if (huf_decode_stream(d->lit, regen, p, left, &d->huf) < 0) return -ZO_E_CORRUPTION;
    } else {
        if (left < 10) return -ZO_E_CORRUPTION;
        size_t s1 = zo_rd16(p), s2 = zo_rd16(p + 2), s3 = zo_rd16(p + 4);
        if (6 + s1 + s2 + s3 > left) return -ZO_E_CORRUPTION;
        size_t s4 = left - 6 - s1 - s2 - s3;
        size_t seg = (regen + 3) / 4;
        if (3 * seg > regen) return -ZO_E_CORRUPTION;
        const uint8_t* q = p + 6;
        if (huf_decode_stream(d->lit, seg, q, s1, &d->huf) < 0) return -ZO_E_CORRUPTION;
        if (huf_decode_stream(d->lit + seg, seg, q + s1, s2, &d->huf) < 0) return -ZO_E_CORRUPTION;
        if (huf_decode_stream(d->lit + 2 * seg, seg, q + s1 + s2, s3, &d->huf) < 0) return -ZO_E_CORRUPTION;
        if (huf_decode_stream(d->lit + 3 * seg, regen - 3 * seg, q + s1 + s2 + s3, s4, &d->huf) < 0) return -ZO_E_CORRUPTION;
    }
    *litSize = regen;
    return (int)(hdr + csize);
}

/* ------------------------------------------------------------------ sequences section */
static int build_seq_table(fse_dtable* dt, int* valid, unsigned mode, unsigned maxSym, unsigned maxLog,
                           const int16_t* defNorm, unsigned defLog, const uint8_t* src, size_t srcSize)
{
    switch (mode) {
    case 0: { int e = fse_build_dtable(dt, defNorm, maxSym, defLog); if (e < 0) return e; *valid = 1; return 0; }
    case 1:
        if (srcSize < 1) return -ZO_E_SRC_SIZE_WRONG;
        if (src[0] > maxSym) return -ZO_E_CORRUPTION;
        fse_build_rle(dt, src[0]); *valid = 1; return 1;
    case 2: {
        int16_t norm[64]; unsigned maxS = maxSym, tl;
        int r = zo_fse_read_ncount(norm, &maxS, &tl, src, srcSize);
        if (r < 0) return -ZO_E_CORRUPTION;
        if (tl > maxLog) return -ZO_E_CORRUPTION;
        int e = fse_build_dtable(dt, norm, maxS, tl); if (e < 0) return -ZO_E_CORRUPTION;
        *valid = 1; return r;
    }
    default:
        if (!*valid) return -ZO_E_CORRUPTION;
        return 0;
    }
}

/* analysis hook (tests/tools/seq_stats.py): when set, every decoded sequence is appended as (litLength, matchLength, offset) */
static uint32_t* g_trace; static size_t g_traceCap, g_traceN;
void zo_set_seq_trace(uint32_t* buf, size_t capTriples) { g_trace = buf; g_traceCap = capTriples; g_traceN = 0; }
size_t zo_seq_trace_count(void) { return g_traceN; }

static int decode_block(zo_dctx* d, uint8_t* ostart, uint8_t* op, uint8_t* oend, const uint8_t* src, size_t srcSize,
                        size_t blockMax, size_t* produced)
{
    size_t litSize = 0;
    uint8_t* const blockStart = op;
    int r = decode_literals(d, src, srcSize, &litSize, blockMax);
    if (r < 0) return r;
    const uint8_t* p = src + r; const uint8_t* end = src + srcSize;
    if (p >= end) return -ZO_E_SRC_SIZE_WRONG;
    unsigned nbSeq = *p++;
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (p + 2 > end) return -ZO_E_SRC_SIZE_WRONG; nbSeq = zo_rd16(p) + 0x7F00; p += 2; }
        else { if (p >= end) return -ZO_E_SRC_SIZE_WRONG; nbSeq = ((nbSeq - 128) << 8) + *p++; }
    }
    const uint8_t* lit = d->lit;
    const uint8_t* litEnd = d->lit + litSize;
    if (nbSeq == 0) {
        if (p != end) return -ZO_E_CORRUPTION;
    } else {
        if (p >= end) return -ZO_E_SRC_SIZE_WRONG;
        unsigned modes = *p++;
        if (modes & 3) return -ZO_E_CORRUPTION;
        r = build_seq_table(&d->ll, &d->llValid, modes >> 6, ZO_MAXLL, ZO_LL_LOGMAX, zo_ll_defnorm, ZO_LL_DEFLOG, p, (size_t)(end - p));
        if (r < 0) return -ZO_E_CORRUPTION;
        p += r;
        r = build_seq_table(&d->of, &d->ofValid, (modes >> 4) & 3, ZO_MAXOFF, ZO_OF_LOGMAX, zo_of_defnorm, ZO_OF_DEFLOG, p, (size_t)(end - p));
        if (r < 0) return -ZO_E_CORRUPTION;
        p += r;
        r = build_seq_table(&d->ml, &d->mlValid, (modes >> 2) & 3, ZO_MAXML, ZO_ML_LOGMAX, zo_ml_defnorm, ZO_ML_DEFLOG, p, (size_t)(end - p));
        if (r < 0) return -ZO_E_CORRUPTION;
        p += r;
        bwd_bits b;
        if (bwd_init(&b, p, (size_t)(end - p)) < 0) return -ZO_E_CORRUPTION;
        unsigned sl = (unsigned)bwd_read(&b, d->ll.log);
        unsigned so = (unsigned)bwd_read(&b, d->of.log);
        unsigned sm = (unsigned)bwd_read(&b, d->ml.log);
        if (b.bits < 0) return -ZO_E_CORRUPTION;
        for (unsigned n = 0; n < nbSeq; n++) {
            fse_cell cl = d->ll.cell[sl], co = d->of.cell[so], cm = d->ml.cell[sm];
            if (co.sym > ZO_MAXOFF) return -ZO_E_CORRUPTION;
            uint64_t ofv = (1ull << co.sym) + bwd_read(&b, co.sym);
            uint32_t ml = zo_ml_base[cm.sym] + (uint32_t)bwd_read(&b, zo_ml_bits[cm.sym]);
            uint32_t ll = zo_ll_base[cl.sym] + (uint32_t)bwd_read(&b, zo_ll_bits[cl.sym]);
            uint64_t offset;
            if (ofv > 3) { offset = ofv - 3; d->rep[2] = d->rep[1]; d->rep[1] = d->rep[0]; d->rep[0] = (uint32_t)offset; }
            else {
                unsigned idx = (unsigned)ofv - 1 + (ll == 0);      /* 0..3 */
                if (idx == 0) offset = d->rep[0];
                else {
                    offset = (idx == 3) ? (uint64_t)d->rep[0] - 1 : d->rep[idx];
                    if (offset == 0) offset = ~(uint64_t)0;        /* zstd.c:46941 (1.5.7): "0 is not valid: input corrupted => force offset to -1 => corruption detected at execSequence" */
                    if (idx != 1) d->rep[2] = d->rep[1];
                    d->rep[1] = d->rep[0]; d->rep[0] = (uint32_t)offset;
                }
            }
            if (n + 1 < nbSeq) {
                sl = cl.base + (unsigned)bwd_read(&b, cl.nbBits);
                sm = cm.base + (unsigned)bwd_read(&b, cm.nbBits);
                so = co.base + (unsigned)bwd_read(&b, co.nbBits);
            }
            if (b.bits < 0) return -ZO_E_CORRUPTION;
            if (g_trace && g_traceN < g_traceCap) { g_trace[3 * g_traceN] = ll; g_trace[3 * g_traceN + 1] = ml; g_trace[3 * g_traceN + 2] = (uint32_t)offset; g_traceN++; }
            /* execute */
            if ((size_t)(litEnd - lit) < ll) return -ZO_E_CORRUPTION;
            if ((size_t)(oend - op) < (size_t)ll + ml) return -ZO_E_DST_TOO_SMALL;
            memcpy(op, lit, ll); op += ll; lit += ll;
            size_t avail = (size_t)(op - ostart);
            if (offset > avail) {
                size_t back = (size_t)offset - avail;                /* reaches into the dictionary */
                if (back > d->dictSize) return -ZO_E_CORRUPTION;
                const uint8_t* dp = d->dict + d->dictSize - back;
                while (ml && back) { *op++ = *dp++; ml--; back--; }
                const uint8_t* m = ostart;
                while (ml) { *op++ = *m++; ml--; }
            } else {
                const uint8_t* m = op - offset;
                while (ml) { *op++ = *m++; ml--; }
            }
            if ((size_t)(op - blockStart) > blockMax) return -ZO_E_CORRUPTION;
        }
        if (b.bits != 0) return -ZO_E_CORRUPTION;
    }
    size_t last = (size_t)(litEnd - lit);
    if ((size_t)(oend - op) < last) return -ZO_E_DST_TOO_SMALL;
    memcpy(op, lit, last); op += last;
    if ((size_t)(op - blockStart) > blockMax) return -ZO_E_CORRUPTION;
    *produced = (size_t)(op - blockStart);
    return 0;
}

/* ------------------------------------------------------------------ frame level */
int zo_get_frame_header(zo_frame_header* h, const void* srcv, size_t srcSize)
{
    const uint8_t* src = (const uint8_t*)srcv;
    memset(h, 0, sizeof(*h));
    if (srcSize < 5) return -ZO_E_SRC_SIZE_WRONG;
    if ((zo_rd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {              /* skippable frame (zstd.c:43706-43715): magic, 4-byte size, that many bytes */
        if (srcSize < 8) return -ZO_E_SRC_SIZE_WRONG;
        h->skippable = 1; h->headerSize = 8; h->windowSize = zo_rd32(src + 4); h->contentSize = 0;   /* ZSTD_getFrameContentSize: 0 (zstd.c:43773) */
        return 0;
    }
    if (zo_rd32(src) != ZO_MAGIC) return -ZO_E_PREFIX_UNKNOWN;
    unsigned fhd = src[4];
    unsigned dictCode = fhd & 3, checksum = (fhd >> 2) & 1, single = (fhd >> 5) & 1, fcsCode = fhd >> 6;
    static const unsigned dictBytes[4] = {0, 1, 2, 4};
    static const unsigned fcsBytes[4] = {0, 2, 4, 8};
    size_t hs = 5 + (single ? 0 : 1) + dictBytes[dictCode] + (fcsCode ? fcsBytes[fcsCode] : (single ? 1 : 0));
    if (fhd & 8) return -ZO_E_FRAMEPARAM_UNSUPPORTED;
    if (srcSize < hs) return -ZO_E_SRC_SIZE_WRONG;
    size_t pos = 5;
    uint64_t windowSize = 0;
    if (!single) {
        unsigned wd = src[pos++];
        unsigned wl = 10 + (wd >> 3);
        if (wl > 31) return -ZO_E_WINDOW_TOO_LARGE;
        windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (wd & 7);
    }
    switch (dictCode) {
    case 1: h->dictID = src[pos]; pos += 1; break;
    case 2: h->dictID = zo_rd16(src + pos); pos += 2; break;
    case 3: h->dictID = zo_rd32(src + pos); pos += 4; break;
    default: break;
    }
    h->contentSize = ZO_CONTENTSIZE_UNKNOWN;
    switch (fcsCode) {
    case 0: if (single) h->contentSize = src[pos]; break;
    case 1: h->contentSize = (uint64_t)zo_rd16(src + pos) + 256; break;
    case 2: h->contentSize = zo_rd32(src + pos); break;
    default: h->contentSize = zo_rd64(src + pos); break;
    }
    if (single) windowSize = h->contentSize;
    h->windowSize = windowSize;
    h->headerSize = (uint32_t)hs;
    h->hasChecksum = checksum;
    h->blockSizeMax = (uint32_t)(windowSize < ZO_BLOCK_MAX ? windowSize : ZO_BLOCK_MAX);
    return 0;
}

uint64_t zo_frame_content_size(const void* src, size_t srcSize)
{
    zo_frame_header h;
    if (zo_get_frame_header(&h, src, srcSize) < 0) return ZO_CONTENTSIZE_ERROR;
    return h.contentSize;
}

int64_t zo_find_frame_compressed_size(const void* srcv, size_t srcSize)
{
    const uint8_t* src = (const uint8_t*)srcv;
    zo_frame_header h; int e = zo_get_frame_header(&h, src, srcSize); if (e < 0) return e;
    if (h.skippable) return 8 + h.windowSize > srcSize ? -ZO_E_SRC_SIZE_WRONG : (int64_t)(8 + h.windowSize);     /* readSkippableFrameSize, zstd.c:43795 */
    size_t pos = h.headerSize;
    for (;;) {
        if (pos + 3 > srcSize) return -ZO_E_SRC_SIZE_WRONG;
        uint32_t bh = zo_rd24(src + pos); pos += 3;
        unsigned type = (bh >> 1) & 3; uint32_t bs = bh >> 3;
        if (type == 3) return -ZO_E_CORRUPTION;
        size_t csz = (type == 1) ? 1 : bs;
        if (pos + csz > srcSize) return -ZO_E_SRC_SIZE_WRONG;
        pos += csz;
        if (bh & 1) break;
    }
    if (h.hasChecksum) { if (pos + 4 > srcSize) return -ZO_E_SRC_SIZE_WRONG; pos += 4; }
    return (int64_t)pos;
}

int64_t zo_decompress_frame(void* dstv, size_t dstCap, const void* srcv, size_t srcSize,
                            const void* dict, size_t dictSize, size_t* srcConsumed)
{
    const uint8_t* src = (const uint8_t*)srcv;
    uint8_t* dst = (uint8_t*)dstv;
    zo_frame_header h; int e = zo_get_frame_header(&h, src, srcSize); if (e < 0) return e;
    if (h.skippable) {              /* ZSTD_decompressStream passes over it and stops at the frame boundary: nothing produced (incomplete: the caller sees a non-zero hint) */
        if (8 + h.windowSize > srcSize) return -ZO_E_SRC_SIZE_WRONG;
        if (srcConsumed) *srcConsumed = 8 + (size_t)h.windowSize;
        return 0;
    }
    zo_dctx* d = (zo_dctx*)calloc(1, sizeof(zo_dctx));
    if (!d) return -ZO_E_MEMORY;
    d->lit = (uint8_t*)malloc(ZO_BLOCK_MAX + 64);
    int64_t result;
    if (!d->lit) { free(d); return -ZO_E_MEMORY; }
    d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
    if (dict && dictSize) {
        e = load_dict(d, (const uint8_t*)dict, dictSize);
        if (e < 0) { result = e; goto done; }
        if (h.dictID && d->dictID != h.dictID) { result = -ZO_E_DICT_WRONG; goto done; }
    }
    {
        size_t pos = h.headerSize;
        uint8_t* op = dst; uint8_t* oend = dst + dstCap;
        /* The reference hands the frame to ZSTD_decompressStream with an output of the declared size (c-ext/decompressor.c:1150). With the
         * content size in the header and room for it libzstd decodes in ONE PASS (zstd.c:44174 ZSTD_decompressFrame): raw and RLE blocks of any
         * size pass, a compressed block above the frame's block maximum is srcSize_wrong (zstd.c:47714). Otherwise it streams
         * (ZSTD_decompressContinue) and every block above the maximum is corruption_detected. */
        const int onePass = h.contentSize != ZO_CONTENTSIZE_UNKNOWN && (uint64_t)dstCap >= h.contentSize;
        for (;;) {
            if (pos + 3 > srcSize) { result = -ZO_E_SRC_SIZE_WRONG; goto done; }
            uint32_t bh = zo_rd24(src + pos); pos += 3;
            unsigned last = bh & 1, type = (bh >> 1) & 3; uint32_t bs = bh >> 3;
            if (type == 3) { result = -ZO_E_CORRUPTION; goto done; }
            if (type == 0) {
                if (pos + bs > srcSize) { result = -ZO_E_SRC_SIZE_WRONG; goto done; }
                if (bs > h.blockSizeMax && !onePass) { result = -ZO_E_CORRUPTION; goto done; }
                if ((size_t)(oend - op) < bs) { result = -ZO_E_DST_TOO_SMALL; goto done; }
                if (bs) memcpy(op, src + pos, bs);
                op += bs; pos += bs;
            } else if (type == 1) {
                if (pos + 1 > srcSize) { result = -ZO_E_SRC_SIZE_WRONG; goto done; }
                if (bs > h.blockSizeMax && !onePass) { result = -ZO_E_CORRUPTION; goto done; }
                if ((size_t)(oend - op) < bs) { result = -ZO_E_DST_TOO_SMALL; goto done; }
                if (bs) memset(op, src[pos], bs);
                op += bs; pos += 1;
            } else {
                if (pos + bs > srcSize) { result = -ZO_E_SRC_SIZE_WRONG; goto done; }
                if (bs > h.blockSizeMax) { result = onePass ? -ZO_E_SRC_SIZE_WRONG : -ZO_E_CORRUPTION; goto done; }
                size_t produced = 0;
                e = decode_block(d, dst, op, oend, src + pos, bs, h.blockSizeMax, &produced);
                if (e < 0) { result = e; goto done; }
                op += produced; pos += bs;
            }
            if (last) break;
        }
        size_t total = (size_t)(op - dst);
        if (h.contentSize != ZO_CONTENTSIZE_UNKNOWN && total != h.contentSize) { result = -ZO_E_CORRUPTION; goto done; }
        if (h.hasChecksum) {
            if (pos + 4 > srcSize) { result = -ZO_E_CHECKSUM_WRONG; goto done; }
            if ((uint32_t)zo_xxh64(dst, total, 0) != zo_rd32(src + pos)) { result = -ZO_E_CHECKSUM_WRONG; goto done; }
            pos += 4;
        }
        if (srcConsumed) *srcConsumed = pos;
        result = (int64_t)total;
    }
done:
    free(d->lit); free(d);
    return result;
}
