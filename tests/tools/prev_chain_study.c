/* tests/tools/prev_chain_study.c -- ANALYSIS AID (test infrastructure; built and run by tests/tools/prev_chain_study.py).
 *
 * Question ("one designed experiment only if it reduces transactions per probed position"): the double-fast
 * search (ZSTD_compressBlock_doubleFast_noDict_generic zstd.c:31039) reads one cell and writes one cell per table per probed
 * position (:31121); with 65 536 frames in flight every one of those is a random HBM transaction, which is what bounds
 * zhip_encode_match_flat_kernel. Alternative formulation: a parallel pre-pass links every position p to the nearest earlier position
 * with the same hash (prevL[p], prevS[p] -- position-ordered arrays, so the serial walk reads them sequentially), the serial walk
 * marks the positions it inserts (a flag in the same word) and answers "what does the table hold for hash(p)" by following the
 * links from p until it meets a marked position: no table, no table writes. What that costs is the number of links followed --
 * each one a random read. This tool measures it: the search runs with its real tables (the truth) and, beside it, the link
 * arrays over ALL positions; for every table lookup it counts how many links the walk needs before it stands on the cell's true
 * content (or on the chain's end for an empty cell).
 *
 * Output (stdout, one line per frame): probes  lookupsL hopsL maxL  lookupsS hopsS maxS  emptyL emptyS  nseq  excess2 over16 over64
 *   (excess2 = link reads beyond the second of a lookup, summed: the rounds a lane would hold its wave back; overN = lookups with more than N reads)
 *   hops = random link reads in total (the first link, prev[p], is the sequential read and not counted; every further candidate
 *   examined is one random read of its flag + link word). */
#include "../../oracle/zo_encode.c"
#include <stdio.h>
#include <stdlib.h>

typedef struct { long probes, lookL, hopL, maxL, lookS, hopS, maxS, emptyL, emptyS, nseq, hist[2][18], excess2, over16, over64; } stats;

static uint32_t* g_prevL; static uint32_t* g_prevS;      /* index space: pos + 2, 0 = none */
static stats* g_st;

static void walk(const uint32_t* prev, uint32_t p, uint32_t truth, int isS)
{
    /* candidates: prev[p], prev[prev[p]], ...; stop at `truth` (a marked position by construction) or at the chain's end */
    long hops = 0; uint32_t q = prev[p];
    while (q != truth && q != 0) { q = prev[q]; hops++; }       /* every step past the first candidate reads that candidate's word */
    if (q != 0) hops++;                                          /* the word of the candidate we stop at (its flag says "marked") */
    if (isS) { g_st->lookS++; g_st->hopS += hops; if (hops > g_st->maxS) g_st->maxS = hops; if (!truth) g_st->emptyS++; }
    else     { g_st->lookL++; g_st->hopL += hops; if (hops > g_st->maxL) g_st->maxL = hops; if (!truth) g_st->emptyL++; }
    if (hops > 2) g_st->excess2 += hops - 2;
    if (hops > 16) g_st->over16++;
    if (hops > 64) g_st->over64++;
    int b = 0; while ((1L << b) <= hops && b < 17) b++;          /* bucket 0: 0 hops, 1: 1, 2: 2-3, 3: 4-7 ... */
    g_st->hist[isS][b]++;
}

/* the oracle's zo_dfast_g with every table READ passed through walk() */
static size_t study_dfast(const uint8_t* src, size_t srcSize, const zo_cpar* cp, uint32_t* hashLong, uint32_t* hashSmall)
{
    const int hl = cp->hlog, hs = cp->clog;
    const int mls = cp->mml <= 4 ? 4 : cp->mml >= 7 ? 7 : cp->mml;
    const uint8_t* const base = src - 2;
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    const uint32_t LOW = 2;
    const uint8_t* anchor = src;
    const uint8_t* ip = src + 1;
    uint32_t off1 = 1, off2 = 0;         /* {1, 4} at a frame's second position: 4 is larger than the history and parked (zstd.c:31091-31098) */
    size_t nseq = 0;
#define RDL(h, p) (walk(g_prevL, (uint32_t)((p) - base), hashLong[h], 0), hashLong[h])
#define RDS(h, p) (walk(g_prevS, (uint32_t)((p) - base), hashSmall[h], 1), hashSmall[h])
    for (;;) {
        size_t step = 1; const uint8_t* nextStep = ip + 256; const uint8_t* ip1 = ip + step;
        size_t mLength; uint32_t offset, curr = 0;
        if (ip1 > ilimit) break;
        uint32_t hl0 = hash_n(ip, hl, 8), idxl0 = RDL(hl0, ip);
        uint32_t hl1 = 0, idxl1 = 0;
        int found = 0;
        do {
            uint32_t hs0 = hash_n(ip, hs, mls), idxs0 = RDS(hs0, ip);
            g_st->probes++;
            curr = (uint32_t)(ip - base);
            hashLong[hl0] = hashSmall[hs0] = curr;
            if (off1 > 0 && zo_rd32(ip + 1 - off1) == zo_rd32(ip + 1)) {
                mLength = common_len(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++; nseq++; found = 1; break;
            }
            hl1 = hash_n(ip1, hl, 8);
            if (idxl0 >= LOW && zo_rd64(base + idxl0) == zo_rd64(ip)) {
                const uint8_t* m = base + idxl0;
                mLength = common_len(ip + 8, m + 8, iend) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > base + LOW && ip[-1] == m[-1]) { ip--; m--; mLength++; }
                found = 2; break;
            }
            idxl1 = RDL(hl1, ip1);
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
            nseq++;
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
                nseq++;
                ip += r; anchor = ip;
            }
        }
    }
    return nseq;
}


/* ---- functional model of the table-free search: the same decisions from the link records alone -------------------------------
 * rec[p] (index space pos + 2): the nearest earlier position with the same long / short hash (over ALL positions) and two flags the
 * search sets when the reference would write that position into the long / short table. A lookup follows the links from p to the
 * first flagged position. One correction keeps "nearest flagged" equal to "last written": the reference writes ip1 into the long
 * table BEFORE curr + 2 (zstd.c:31213 then :31227); when ip1 == curr + 3 and both hash alike the cell ends up holding curr + 2, so
 * ip1 is not flagged then. Every other write order is ascending in position. */
typedef struct { uint32_t prevL, prevS; uint8_t fl; } linkrec;
static linkrec* g_rec;
static uint32_t look(uint32_t p, int isS)
{
    uint32_t q = isS ? g_rec[p].prevS : g_rec[p].prevL;
    while (q && !(g_rec[q].fl & (isS ? 2 : 1))) q = isS ? g_rec[q].prevS : g_rec[q].prevL;
    return q;
}
static size_t links_dfast(zo_seq* seqs, const uint8_t* src, size_t srcSize)
{
    const uint8_t* const base = src - 2;
    const uint8_t* const iend = src + srcSize;
    const uint8_t* const ilimit = iend - 8;
    const uint32_t LOW = 2;
    const uint8_t* anchor = src;
    const uint8_t* ip = src + 1;
    uint32_t off1 = 1, off2 = 0;
    size_t nseq = 0;
#define IDX(p) ((uint32_t)((p) - base))
#define SEQ(LL, OB, ML) do { seqs[nseq].litLength = (uint32_t)(LL); seqs[nseq].offBase = (OB); seqs[nseq].matchLength = (uint32_t)(ML); nseq++; } while (0)
    for (;;) {
        size_t step = 1; const uint8_t* nextStep = ip + 256; const uint8_t* ip1 = ip + step;
        size_t mLength; uint32_t offset = 0, curr = 0;
        if (ip1 > ilimit) break;
        uint32_t idxl0 = look(IDX(ip), 0), idxl1 = 0;
        int found = 0;
        do {
            uint32_t idxs0 = look(IDX(ip), 1);
            curr = IDX(ip);
            g_rec[curr].fl |= 3;
            if (off1 > 0 && zo_rd32(ip + 1 - off1) == zo_rd32(ip + 1)) {
                mLength = common_len(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++; SEQ(ip - anchor, 1, mLength); found = 1; break;
            }
            if (idxl0 >= LOW && zo_rd64(base + idxl0) == zo_rd64(ip)) {
                const uint8_t* m = base + idxl0;
                mLength = common_len(ip + 8, m + 8, iend) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > base + LOW && ip[-1] == m[-1]) { ip--; m--; mLength++; }
                found = 2; break;
            }
            idxl1 = look(IDX(ip1), 0);
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
            idxl0 = idxl1;
        } while (ip1 <= ilimit);
        if (!found) break;
        const uint32_t i1 = IDX(ip1);
        int flag1 = 0;
        if (found == 2) {
            off2 = off1; off1 = offset;
            flag1 = step < 4;
            SEQ(ip - anchor, offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = curr + 2;
            if (getenv("NOQUIRK") == NULL && flag1 && i1 == curr + 3 && g_rec[i1].prevL == ins) flag1 = 0;      /* overwritten by curr + 2 right after */
            if (flag1) g_rec[i1].fl |= 1;
            g_rec[ins].fl |= 3;
            g_rec[IDX(ip - 2)].fl |= 1;
            g_rec[IDX(ip - 1)].fl |= 2;
            while (ip <= ilimit && off2 > 0 && zo_rd32(ip) == zo_rd32(ip - off2)) {
                size_t r = common_len(ip + 4, ip + 4 - off2, iend) + 4;
                uint32_t t = off2; off2 = off1; off1 = t;
                g_rec[IDX(ip)].fl |= 3;
                SEQ(0, 1, r);
                ip += r; anchor = ip;
            }
        } else if (flag1) g_rec[i1].fl |= 1;
    }
    return nseq;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s sources.bin frameSize\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb"); const size_t n = (size_t)atol(argv[2]);
    uint8_t* buf = (uint8_t*)malloc(n + 64);
    zo_cpar cp; zo_get_cparams(&cp, 3, n);
    const int mls = cp.mml <= 4 ? 4 : cp.mml >= 7 ? 7 : cp.mml;
    uint32_t* hashLong = (uint32_t*)malloc(sizeof(uint32_t) << cp.hlog);
    uint32_t* hashSmall = (uint32_t*)malloc(sizeof(uint32_t) << cp.clog);
    g_prevL = (uint32_t*)malloc(4 * (n + 16)); g_prevS = (uint32_t*)malloc(4 * (n + 16));
    fprintf(stderr, "cparams: wlog %d hashLog %d chainLog %d minMatch %d\n", cp.wlog, cp.hlog, cp.clog, cp.mml);
    stats tot; memset(&tot, 0, sizeof tot);
    while (fread(buf, 1, n, f) == n) {
        memset(buf + n, 0, 64);
        /* pre-pass: links over ALL positions that can be hashed (8 bytes readable), in position order */
        memset(hashLong, 0, sizeof(uint32_t) << cp.hlog); memset(hashSmall, 0, sizeof(uint32_t) << cp.clog);
        memset(g_prevL, 0, 4 * (n + 16)); memset(g_prevS, 0, 4 * (n + 16));
        for (size_t p = 0; p + 8 <= n; p++) {
            uint32_t a = hash_n(buf + p, cp.hlog, 8), b = hash_n(buf + p, cp.clog, mls);
            g_prevL[p + 2] = hashLong[a]; hashLong[a] = (uint32_t)p + 2;
            g_prevS[p + 2] = hashSmall[b]; hashSmall[b] = (uint32_t)p + 2;
        }
        memset(hashLong, 0, sizeof(uint32_t) << cp.hlog); memset(hashSmall, 0, sizeof(uint32_t) << cp.clog);
        stats st; memset(&st, 0, sizeof st); g_st = &st;
        st.nseq = (long)study_dfast(buf, n, &cp, hashLong, hashSmall);
        {   /* the model against the oracle's search: same sequences or abort */
            static zo_seq* sa; static zo_seq* sb; static uint8_t* lits;
            if (!sa) { sa = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 64)); sb = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 64)); lits = (uint8_t*)malloc(n + 64); g_rec = (linkrec*)malloc(sizeof(linkrec) * (n + 16)); }
            for (size_t p = 0; p < n + 16; p++) { g_rec[p].prevL = g_prevL[p]; g_rec[p].prevS = g_prevS[p]; g_rec[p].fl = 0; }
            size_t litSize = 0;
            const size_t na = zo_dfast(sa, lits, &litSize, buf, n, &cp, hashLong, hashSmall);
            const size_t nb = links_dfast(sb, buf, n);
            if (na != nb || memcmp(sa, sb, na * sizeof(zo_seq))) { fprintf(stderr, "MODEL MISMATCH: %zu vs %zu sequences\n", na, nb); return 1; }
        }
        printf("%ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld\n", st.probes, st.lookL, st.hopL, st.maxL, st.lookS, st.hopS, st.maxS, st.emptyL, st.emptyS, st.nseq, st.excess2, st.over16, st.over64);
        tot.probes += st.probes; tot.lookL += st.lookL; tot.hopL += st.hopL; tot.lookS += st.lookS; tot.hopS += st.hopS;
        tot.emptyL += st.emptyL; tot.emptyS += st.emptyS; tot.nseq += st.nseq;
        for (int t = 0; t < 2; t++) for (int b = 0; b < 18; b++) tot.hist[t][b] += st.hist[t][b];
    }
    for (int t = 0; t < 2; t++) {
        fprintf(stderr, "%s table, lookups by number of random link reads (0, 1, 2-3, 4-7, ...):", t ? "short" : "long");
        for (int b = 0; b < 18; b++) fprintf(stderr, " %ld", tot.hist[t][b]);
        fprintf(stderr, "\n");
    }
    return 0;
}
