/* tests/tools/tans_sync_study.c -- ANALYSIS AID (test infrastructure; built and run by tests/tools/tans_sync_study.py).
 *
 * Question: a zstd sequences bitstream carries no restart markers -- three interleaved tANS states (LL, OF, ML)
 * plus raw extra bits, read backwards (ZSTD_decodeSequence zstd.c:46862, BIT_reloadDStream :2517). If a second decoder is started at
 * an ARBITRARY bit position with ARBITRARY states, after how many sequences does its (bit position, three states) coincide with
 * the true decoder's trajectory? If that number is small against a frame's ~9 000 sequences, a frame's chain can be cut into pieces
 * decoded by different lane groups (self-synchronising decoding, as established for prefix codes); if not, K2 stays a serial chain.
 *
 * Method: for every frame, decode the true trajectory and remember, per bit position that is a step boundary, the step index and
 * the three states there. Then, for `trials` start positions spread evenly over the bitstream, start a decoder there with each of
 * a few arbitrary state triples (all-zero; pseudo-random) and run it until it lands on a true step boundary with the true states
 * (synchronised for good from there: same position + same states = same future) or has run `maxSteps` steps or leaves the stream.
 * Output: one line per trial "steps_to_sync" (or -1), summarised by the python driver.
 *
 * The oracle's own table builders are reused by including its source (static functions); nothing here is product code. */
#include "../../oracle/zo_decode.c"
#include <stdio.h>
#include <stdlib.h>

typedef struct { int32_t step; uint16_t sl, so, sm; } boundary;

static uint32_t rnd_state = 12345;
static uint32_t rnd(void) { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

/* one block's sequences section: p..end is what follows the literals section */
static void study_block(zo_dctx* d, const uint8_t* p, const uint8_t* end, int trials, int maxSteps, FILE* out, long* nSeqTotal)
{
    if (p >= end) return;
    unsigned nbSeq = *p++;
    if (nbSeq > 127) {
        if (nbSeq == 255) { nbSeq = zo_rd16(p) + 0x7F00; p += 2; }
        else nbSeq = ((nbSeq - 128) << 8) + *p++;
    }
    if (nbSeq < 512) return;                               /* tiny blocks: nothing to parallelise */
    unsigned modes = *p++;
    int r = build_seq_table(&d->ll, &d->llValid, modes >> 6, ZO_MAXLL, ZO_LL_LOGMAX, zo_ll_defnorm, ZO_LL_DEFLOG, p, (size_t)(end - p));
    if (r < 0) return; p += r;
    r = build_seq_table(&d->of, &d->ofValid, (modes >> 4) & 3, ZO_MAXOFF, ZO_OF_LOGMAX, zo_of_defnorm, ZO_OF_DEFLOG, p, (size_t)(end - p));
    if (r < 0) return; p += r;
    r = build_seq_table(&d->ml, &d->mlValid, (modes >> 2) & 3, ZO_MAXML, ZO_ML_LOGMAX, zo_ml_defnorm, ZO_ML_DEFLOG, p, (size_t)(end - p));
    if (r < 0) return; p += r;
    bwd_bits b;
    if (bwd_init(&b, p, (size_t)(end - p)) < 0) return;
    const int64_t totalBits = b.bits;
    boundary* at = (boundary*)malloc(sizeof(boundary) * (size_t)(totalBits + 1));
    for (int64_t i = 0; i <= totalBits; i++) at[i].step = -1;
    unsigned sl = (unsigned)bwd_read(&b, d->ll.log), so = (unsigned)bwd_read(&b, d->of.log), sm = (unsigned)bwd_read(&b, d->ml.log);
    for (unsigned n = 0; n < nbSeq; n++) {
        at[b.bits].step = (int32_t)n; at[b.bits].sl = (uint16_t)sl; at[b.bits].so = (uint16_t)so; at[b.bits].sm = (uint16_t)sm;
        fse_cell cl = d->ll.cell[sl], co = d->of.cell[so], cm = d->ml.cell[sm];
        b.bits -= co.sym; b.bits -= zo_ml_bits[cm.sym]; b.bits -= zo_ll_bits[cl.sym];
        if (n + 1 < nbSeq) {
            sl = cl.base + (unsigned)bwd_read(&b, cl.nbBits);
            sm = cm.base + (unsigned)bwd_read(&b, cm.nbBits);
            so = co.base + (unsigned)bwd_read(&b, co.nbBits);
        }
        if (b.bits < 0) { free(at); return; }
    }
    *nSeqTotal += nbSeq;
    const unsigned ml = 1u << d->ll.log, mo = 1u << d->of.log, mm = 1u << d->ml.log;
    for (int t = 0; t < trials; t++) {
        /* start positions: evenly spread cut points of the stream, like a kernel would choose them (it knows only the bit length) */
        const int64_t start = totalBits - (totalBits * (int64_t)(t + 1)) / (trials + 1);
        for (int variant = 0; variant < 2; variant++) {
            bwd_bits s; s.p = b.p; s.bits = start;
            unsigned xl = variant ? rnd() % ml : 0, xo = variant ? rnd() % mo : 0, xm = variant ? rnd() % mm : 0;
            int steps = 0, synced = -1; int32_t trueStepAtSync = -1;
            while (steps < maxSteps) {
                if (s.bits >= 0 && at[s.bits].step >= 0 && at[s.bits].sl == xl && at[s.bits].so == xo && at[s.bits].sm == xm) { synced = steps; trueStepAtSync = at[s.bits].step; break; }
                fse_cell cl = d->ll.cell[xl], co = d->of.cell[xo], cm = d->ml.cell[xm];
                unsigned osym = co.sym > 31 ? 31 : co.sym;
                s.bits -= osym; s.bits -= zo_ml_bits[cm.sym]; s.bits -= zo_ll_bits[cl.sym];
                if (s.bits < 32) break;
                xl = cl.base + (unsigned)bwd_read(&s, cl.nbBits);
                xm = cm.base + (unsigned)bwd_read(&s, cm.nbBits);
                xo = co.base + (unsigned)bwd_read(&s, co.nbBits);
                steps++;
            }
            /* also: how far (in true steps) is the sync point from the true step nearest the start position */
            fprintf(out, "%d %d %u %lld\n", synced, trueStepAtSync, nbSeq, (long long)totalBits);
        }
    }
    free(at);
}

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s frames.bin out.txt trials maxSteps\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb"); FILE* out = fopen(argv[2], "w");
    const int trials = atoi(argv[3]), maxSteps = atoi(argv[4]);
    zo_dctx* d = (zo_dctx*)calloc(1, sizeof(zo_dctx));
    uint8_t* litBuf = (uint8_t*)malloc(ZO_BLOCK_MAX + 64);
    uint32_t len; long nFrames = 0, nSeq = 0;
    uint8_t* buf = (uint8_t*)malloc(1 << 20);
    while (fread(&len, 4, 1, f) == 1) {
        if (fread(buf, 1, len, f) != len) break;
        zo_frame_header h;
        if (zo_get_frame_header(&h, buf, len) < 0) continue;
        size_t pos = h.headerSize;
        memset(d, 0, sizeof(*d));
        d->lit = litBuf;
        d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
        for (;;) {
            if (pos + 3 > len) break;
            const uint32_t bh = buf[pos] | (buf[pos + 1] << 8) | ((uint32_t)buf[pos + 2] << 16);
            const uint32_t last = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
            pos += 3;
            if (type == 2) {
                size_t litSize = 0;
                int r = decode_literals(d, buf + pos, bs, &litSize, 131072);
                if (r >= 0) study_block(d, buf + pos + r, buf + pos + bs, trials, maxSteps, out, &nSeq);
                pos += bs;
            } else pos += type == 1 ? 1 : bs;
            if (last) break;
        }
        nFrames++;
    }
    fprintf(stderr, "frames %ld, sequences in studied blocks %ld\n", nFrames, nSeq);
    fclose(out); fclose(f);
    return 0;
}
