// zhip_decode_pipeline.hpp -- phase-split batch decoder for the common case (dictionary-less frames made of ONE block,
// which is every frame multi_compress_to_buffer produces for inputs <= 128 KiB). Profile-driven (profiles/r01a): the
// fused one-wave-per-frame kernel spends 63 % of its cycles in the single-lane tANS chain and is limited to ~8-11 frames
// per CU by its 14 KiB of LDS. Splitting the frame loop by phase lets each phase use the mapping that fits it:
//
//   K1 zhip_decode_lit_kernel   one wave per frame : header + block header + literals section (Huffman, 4 lanes / 4 streams)
//                                                     -> per-frame literal slot; raw / RLE single-block frames finish here
//   K2 zhip_decode_seq_kernel   one LANE per frame : FSE tables (2-byte cells, 2.8 KiB of LDS per frame) + the serial tANS
//                                                     decode, 16 frames per wave, up to 48 frames in flight per CU;
//                                                     emits packed 8-byte sequences to HBM
//   K3 zhip_decode_exec_kernel  one wave per frame : reads 64 sequences per batch (coalesced), assembles the batch output in
//                                                     LDS, flushes with 16-byte stores
//
// Anything else (multi-block frames, dictionaries, oversize offsets) is routed to the generic fused kernel through a
// fallback list, so results are identical on every input.
#pragma once
#include "zhip_decode_kernel.hpp"

// ------------------------------------------------------------------------------------------ K1
ZH_DEVFN void zp_lit_body(const ZhipPipeArgs& a, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counters + 0, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        const uint32_t i = zh_first(L.misc[7]);
        zh_sync();
        if (i >= a.count) break;
        const uint32_t f = a.first + i;
        ZdMeta m;
        m.status = 0; m.path = 0; m.seqOff = m.seqEnd = 0; m.litSize = 0; m.litMode = 0; m.litOff = 0; m.nbSeq = 0;
        m.blockMax = 0; m.fcsLo = m.fcsHi = 0xFFFFFFFFu; m.produced = 0; m.hasChecksum = 0; m.checksum = 0;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
        uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
        const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
        int err = 0; bool fallback = false;
        do {
            if (srcSize64 > 0x7FFFFFFFull) { fallback = true; break; }
            const uint32_t srcSize = (uint32_t)srcSize64;
            const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
            if (srcSize < 5) { err = ZE_SRC_SIZE_WRONG; break; }
            if (zh_ld32(src) != ZF_MAGIC) { err = ZE_PREFIX_UNKNOWN; break; }
            const uint32_t fhd = src[4];
            const uint32_t dictCode = fhd & 3, single = (fhd >> 5) & 1, fcsCode = fhd >> 6, hasChecksum = (fhd >> 2) & 1;
            const uint32_t dictBytes = dictCode == 3 ? 4 : dictCode;
            const uint32_t fcsBytes = fcsCode == 0 ? single : (1u << fcsCode);
            const uint32_t hs = 5 + (single ? 0 : 1) + dictBytes + fcsBytes;
            if (fhd & 8) { err = ZE_FRAMEPARAM_UNSUPPORTED; break; }
            if (srcSize < hs) { err = ZE_SRC_SIZE_WRONG; break; }
            if (dictCode) { fallback = true; break; }                       // dictionary frames: generic kernel decides
            uint32_t pos = 5;
            uint64_t windowSize = 0;
            if (!single) {
                const uint32_t wd = src[pos++], wl = 10 + (wd >> 3);
                if (wl > 31) { err = ZE_WINDOW_TOO_LARGE; break; }
                windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (wd & 7);
            }
            uint64_t fcs = ~0ull;
            if (fcsCode == 0) { if (single) fcs = src[pos]; }
            else if (fcsCode == 1) fcs = (uint64_t)zh_ld16(src + pos) + 256;
            else if (fcsCode == 2) fcs = zh_ld32(src + pos);
            else fcs = zh_ld64(src + pos);
            pos += fcsBytes;
            if (single) windowSize = fcs;
            if (windowSize > a.maxWindowSize) { err = ZE_WINDOW_TOO_LARGE; break; }
            const uint32_t blockMax = windowSize < ZF_BLOCK_MAX ? (uint32_t)windowSize : ZF_BLOCK_MAX;
            m.blockMax = blockMax; m.fcsLo = (uint32_t)fcs; m.fcsHi = (uint32_t)(fcs >> 32);
            if (pos + 3 > srcSize) { err = ZE_SRC_SIZE_WRONG; break; }
            const uint32_t bh = zh_ld24(src + pos); pos += 3;
            const uint32_t lastBlock = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
            if (!lastBlock) { fallback = true; break; }                      // multi-block frame
            if (type == 3) { err = ZE_CORRUPTION; break; }
            if (type < 2) {                                                 // one raw / RLE block: finish right here
                if (type == 0 ? pos + bs > srcSize : pos + 1 > srcSize) { err = ZE_SRC_SIZE_WRONG; break; }
                if (bs > blockMax) { err = ZE_CORRUPTION; break; }
                if (bs > cap) { err = ZE_DST_TOO_SMALL; break; }
                if (type == 0) zd_copy_wave(dst, src + pos, bs); else zd_fill_wave(dst, src[pos], bs);
                if (fcs != ~0ull && fcs != bs) { err = ZE_CORRUPTION; break; }
                m.produced = bs;
                if (hasChecksum) {
                    const uint32_t cpos = pos + (type == 0 ? bs : 1);
                    if (cpos + 4 > srcSize) { err = ZE_CHECKSUM_WRONG; break; }
                    zd_fence();
                    zh_sync();
                    if (zh_opaque(lane) == 0) L.misc[0] = (uint32_t)ze_xxh64(dst, bs);
                    zh_sync();
                    const uint32_t digest = zh_first(L.misc[0]);
                    zh_sync();
                    if (digest != zh_ld32(src + cpos)) { err = ZE_CHECKSUM_WRONG; break; }
                }
                break;
            }
            if (pos + bs > srcSize) { err = ZE_SRC_SIZE_WRONG; break; }
            if (bs > ZF_BLOCK_MAX || bs < 2) { err = ZE_CORRUPTION; break; }
            ZdState st;
            st.rep0 = 1; st.rep1 = 4; st.rep2 = 8; st.hufCount = 0; st.llLog = st.ofLog = st.mlLog = 0xFF;
            uint8_t* lit = a.litArena + (size_t)i * ZP_LIT_STRIDE;
            st.litPtr = lit; st.litSize = 0; st.litRLE = 0; st.rleByte = 0;
            ZdProf P; P.on = false;
            const int r = zd_literals(L, st, src + pos, bs, lit, blockMax, P);
            if (r < 0) { err = -r; break; }
            m.litSize = st.litSize;
            if (st.litRLE) { m.litMode = 2; m.litOff = st.rleByte; }
            else if (st.litPtr == lit) { m.litMode = 1; m.litOff = 0; }
            else { m.litMode = 0; m.litOff = (uint32_t)(st.litPtr - src); }
            m.seqOff = pos + (uint32_t)r; m.seqEnd = pos + bs;
            if (hasChecksum) {
                if (pos + bs + 4 > srcSize) { err = ZE_CHECKSUM_WRONG; break; }
                m.hasChecksum = 1; m.checksum = zh_ld32(src + pos + bs);
            }
            m.path = 1;
        } while (false);
        if (fallback) { m.path = 2; }
        if (err) { m.status = err; m.path = 0; }
        zh_sync();
        if (zh_opaque(lane) == 0) {
            a.meta[i] = m;
            if (m.path == 2) { const uint32_t k = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[k] = f; }
            if (m.path == 0) { a.status[f] = m.status; a.outSizes[f] = m.status ? 0 : m.produced; }
        }
        zd_fence();
    }
}

// ------------------------------------------------------------------------------------------ K2 (one lane == one frame)
// 2-byte FSE cell: symbol[10:16) | x[0:10) where x = normalized count + rank; nbBits = log - highbit(x); base = (x << nbBits) - size
struct ZpLaneLDS { uint16_t ll[512]; uint16_t ml[512]; uint16_t of[256]; int16_t norm[64]; uint16_t next[64]; uint8_t pad[4]; };

// per-lane forward-bit NCount reader (same format logic as zd_read_ncount, private arrays)
ZH_DEVFN int zp_read_ncount(int16_t* norm, const uint8_t* src, const uint8_t* end, uint32_t* pMax, uint32_t* pLog)
{
    if (src >= end) return -ZE_SRC_SIZE_WRONG;
    uint32_t bitpos = 0;
    const uint32_t srcBytes = (uint32_t)(end - src);
#define ZP_PEEK(n) ((uint32_t)((zd_ld64_bounded(src + (bitpos >> 3), end) >> (bitpos & 7)) & ((1ull << (n)) - 1)))
    const int al = (int)ZP_PEEK(4) + 5; bitpos += 4;
    if (al > 9) return -ZE_TABLELOG_TOO_LARGE;
    *pLog = (uint32_t)al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbBits = al + 1;
    uint32_t sym = 0; const uint32_t maxS = *pMax;
    int prev0 = 0;
    while (remaining > 1 && sym <= maxS) {
        if (prev0) {
            for (;;) {
                const uint32_t r = ZP_PEEK(2); bitpos += 2;
                for (uint32_t k = 0; k < r && sym <= maxS; k++) norm[sym++] = 0;
                if (r != 3) break;
                if (bitpos > srcBytes * 8) return -ZE_CORRUPTION;
            }
            if (sym > maxS) return -ZE_MAXSYMBOL_TOO_SMALL;
        }
        const int max = (2 * threshold - 1) - remaining;
        int count;
        const int low = (int)ZP_PEEK((uint32_t)nbBits - 1);
        if (low < max) { count = low; bitpos += (uint32_t)nbBits - 1; }
        else { count = (int)ZP_PEEK((uint32_t)nbBits); if (count >= threshold) count -= max; bitpos += (uint32_t)nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        if (remaining < 1) return -ZE_CORRUPTION;
        norm[sym++] = (int16_t)count;
        prev0 = (count == 0);
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if (((bitpos + 7) >> 3) > srcBytes) return -ZE_CORRUPTION;
    }
#undef ZP_PEEK
    if (remaining != 1) return -ZE_CORRUPTION;
    *pMax = sym - 1;
    return (int)((bitpos + 7) >> 3);
}

// serial table construction by one lane (RFC 8878 4.1.1): spread, then number each symbol's cells in table order
ZH_DEVFN int zp_build_table(uint16_t* cells, const int16_t* norm, uint16_t* next, uint32_t maxSym, uint32_t lg)
{
    const uint32_t S = 1u << lg, mask = S - 1, step = (S >> 1) + (S >> 3) + 3;
    uint32_t high = S - 1, total = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) { cells[high--] = (uint16_t)s; next[s] = 1; total += 1; }
        else { next[s] = (uint16_t)norm[s]; total += (uint32_t)norm[s]; }
    }
    if (total != S) return -ZE_CORRUPTION;
    uint32_t pos = 0;
    for (uint32_t s = 0; s <= maxSym; s++)
        for (int i = 0; i < norm[s]; i++) {
            cells[pos] = (uint16_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    if (pos != 0) return -ZE_CORRUPTION;
    for (uint32_t u = 0; u < S; u++) { const uint32_t s = cells[u]; cells[u] = (uint16_t)((s << 10) | next[s]++); }
    return 0;
}

ZH_DEVFN int zp_seq_table(uint16_t* cells, ZpLaneLDS* Ll, uint32_t mode, int kind, uint32_t* pLog, const uint8_t* p, const uint8_t* end)
{
    const uint32_t maxSym = kind == ZD_KIND_LL ? ZF_MAXLL : kind == ZD_KIND_ML ? ZF_MAXML : ZF_MAXOFF;
    const uint32_t maxLog = kind == ZD_KIND_OF ? ZF_OF_LOGMAX : ZF_LL_LOGMAX;
    if (mode == 0) {
        const int16_t* def = kind == ZD_KIND_LL ? zc_llDef : kind == ZD_KIND_ML ? zc_mlDef : zc_ofDef;
        const uint32_t ms = kind == ZD_KIND_LL ? 35 : kind == ZD_KIND_ML ? 52 : 28, lg = kind == ZD_KIND_OF ? 5 : 6;
        for (uint32_t s = 0; s <= ms; s++) Ll->norm[s] = def[s];
        if (zp_build_table(cells, Ll->norm, Ll->next, ms, lg) < 0) return -ZE_CORRUPTION;
        *pLog = lg; return 0;
    }
    if (mode == 1) {
        if (p >= end) return -ZE_SRC_SIZE_WRONG;
        const uint32_t s = p[0];
        if (s > maxSym) return -ZE_CORRUPTION;
        cells[0] = (uint16_t)((s << 10) | 1);          // x = 1: nbBits = 0 - 0, base = 1 - 1 = 0
        *pLog = 0; return 1;
    }
    if (mode == 2) {
        uint32_t ms = maxSym, lg = 0;
        const int r = zp_read_ncount(Ll->norm, p, end, &ms, &lg);
        if (r < 0 || lg > maxLog) return -ZE_CORRUPTION;
        if (zp_build_table(cells, Ll->norm, Ll->next, ms, lg) < 0) return -ZE_CORRUPTION;
        *pLog = lg; return r;
    }
    return -ZE_CORRUPTION;                              // "repeat" has nothing to repeat in a first block
}

// the whole sequences section of one frame, decoded by ONE lane straight from global memory
ZH_DEVFN int zp_decode_sequences(const uint8_t* p, const uint8_t* end, ZpLaneLDS* Ll, const uint32_t* llBase, const uint32_t* mlBase,
                                 const uint8_t* llBits, const uint8_t* mlBits, uint64_t* out, uint32_t* pNbSeq)
{
    *pNbSeq = 0;
    if (p >= end) return ZE_SRC_SIZE_WRONG;
    uint32_t nbSeq = *p++;
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (p + 2 > end) return ZE_SRC_SIZE_WRONG; nbSeq = zh_ld16(p) + 0x7F00; p += 2; }
        else { if (p >= end) return ZE_SRC_SIZE_WRONG; nbSeq = ((nbSeq - 128) << 8) + *p++; }
    }
    if (nbSeq == 0) return p == end ? 0 : ZE_CORRUPTION;
    if (nbSeq > ZP_SEQ_CAP) return ZE_CORRUPTION;
    if (p >= end) return ZE_SRC_SIZE_WRONG;
    const uint32_t modes = *p++;
    if (modes & 3) return ZE_CORRUPTION;
    uint32_t llLog = 0, ofLog = 0, mlLog = 0;
    int r = zp_seq_table(Ll->ll, Ll, modes >> 6, ZD_KIND_LL, &llLog, p, end); if (r < 0) return -r; p += r;
    r = zp_seq_table(Ll->of, Ll, (modes >> 4) & 3, ZD_KIND_OF, &ofLog, p, end); if (r < 0) return -r; p += r;
    r = zp_seq_table(Ll->ml, Ll, (modes >> 2) & 3, ZD_KIND_ML, &mlLog, p, end); if (r < 0) return -r; p += r;
    if (p >= end) return ZE_CORRUPTION;
    ZdPBits b;
    if (!zd_pb_init(b, p, (uint32_t)(end - p))) return ZE_CORRUPTION;
    int32_t left = ((end - p) >= 8) ? (int32_t)(end - p) * 8 - (int32_t)b.used : 64 - (int32_t)b.used;
    // The loop below is the hot serial chain of the whole decoder, so it is written branch-lean: reads of n in [0,32]
    // bits never branch on n, renormalisation is a funnel shift of (c, d) with the next load already issued, extra-bit
    // counts come from nibble tables instead of a second dependent LDS lookup, repcode handling is select-based, and the
    // rare sequence with more than 31 extra bits takes a side path.
    const uint64_t kLL = 0xCBA9876433221111ull;    // LL_bits[16..31] = 1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12
    const uint64_t kML = 0xBA98754433221111ull;    // ML_bits[32..47] = 1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11
    const uint8_t* const start = b.start;
    const uint8_t* ptr = b.ptr;
    uint64_t c = b.c, d = b.d;
    uint32_t used = b.used;
    const bool tiny = (end - p) < 8;                 // whole stream already sits in c; d stays 0
#define ZP_PEEK(n) ((uint32_t)(((c << (used & 63)) >> 1) >> (63 - (n))))
#define ZP_NORM() do { uint32_t nb_ = used >> 3; if (nb_ > 7) nb_ = 7; const uint32_t room_ = (uint32_t)(ptr - start); if (nb_ > room_) nb_ = room_; \
        c = (c << (8 * nb_)) | ((d >> 1) >> (63 - 8 * nb_)); ptr -= nb_; used -= 8 * nb_; \
        const uint32_t r2_ = room_ - nb_; \
        if (r2_ >= 8) d = zh_ld64(ptr - 8); else d = (tiny || r2_ == 0) ? 0ull : (zh_ld64(start) << (8 * (8 - r2_))); } while (0)
    uint32_t sL, sO, sM;
    ZP_NORM();
    sL = ZP_PEEK(llLog); used += llLog; sO = ZP_PEEK(ofLog); used += ofLog; sM = ZP_PEEK(mlLog); used += mlLog;
    left -= (int32_t)(llLog + ofLog + mlLog);
    ZP_NORM();
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8, bad = 0;
    const uint32_t sizeL = 1u << llLog, sizeO = 1u << ofLog, sizeM = 1u << mlLog;
    for (uint32_t n = 0; n < nbSeq; n++) {
        const uint32_t cL = Ll->ll[sL], cM = Ll->ml[sM], cO = Ll->of[sO];
        const uint32_t symL = cL >> 10, symM = cM >> 10, symO = cO >> 10;
        const uint32_t nibL = (uint32_t)(kLL >> ((4 * symL) & 63)) & 15u, nibM = (uint32_t)(kML >> ((4 * symM) & 63)) & 15u;
        const uint32_t bitsL = symL < 16 ? 0u : (symL < 32 ? nibL : symL - 19);
        const uint32_t bitsM = symM < 32 ? 0u : (symM < 48 ? nibM : symM - 36);
        const uint32_t extra = symO + bitsM + bitsL;
        uint32_t xo, xm, xl;
        if (extra > 31) {                               // rare: long offsets plus long lengths
            xo = ZP_PEEK(symO); used += symO; ZP_NORM();
            xm = ZP_PEEK(bitsM); used += bitsM; xl = ZP_PEEK(bitsL); used += bitsL; ZP_NORM();
        } else {
            xo = ZP_PEEK(symO); used += symO; xm = ZP_PEEK(bitsM); used += bitsM; xl = ZP_PEEK(bitsL); used += bitsL;
        }
        const uint32_t ofv = (1u << symO) + xo;
        const uint32_t mlv = mlBase[symM] + xm;
        const uint32_t llv = llBase[symL] + xl;
        // repcode resolution (RFC 8878 3.1.1.5) with selects
        const uint32_t idx = ofv - 1 + (llv == 0);           // meaningful when ofv <= 3
        uint32_t ro = idx == 0 ? rep0 : idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
        if (ro == 0) ro = 1;
        const bool isRep = ofv <= 3;
        const uint32_t offset = isRep ? ro : ofv - 3;
        const bool shift3 = !isRep || idx >= 2;               // rep2 <- rep1
        const bool shift2 = !isRep || idx >= 1;               // rep1 <- rep0, rep0 <- offset
        rep2 = shift3 ? rep1 : rep2;
        rep1 = shift2 ? rep0 : rep1;
        rep0 = shift2 ? offset : rep0;
        // state updates (skipped bit-wise for the last sequence by zeroing the widths)
        const uint32_t live = n + 1 < nbSeq;
        const uint32_t xL = cL & 1023, xM = cM & 1023, xO = cO & 1023;
        const uint32_t nbL = live ? llLog - (uint32_t)zh_highbit32(xL) : 0u;
        const uint32_t nbM = live ? mlLog - (uint32_t)zh_highbit32(xM) : 0u;
        const uint32_t nbO = live ? ofLog - (uint32_t)zh_highbit32(xO) : 0u;
        const uint32_t tL = ZP_PEEK(nbL); used += nbL;
        const uint32_t tM = ZP_PEEK(nbM); used += nbM;
        const uint32_t tO = ZP_PEEK(nbO); used += nbO;
        sL = ((xL << nbL) - sizeL + tL) & (sizeL - 1);
        sM = ((xM << nbM) - sizeM + tM) & (sizeM - 1);
        sO = ((xO << nbO) - sizeO + tO) & (sizeO - 1);
        left -= (int32_t)(extra + nbL + nbM + nbO);
        ZP_NORM();                                       // the load it issues is consumed one iteration later
        bad |= (uint32_t)(left < 0) | ((offset >> 30) << 1);
        out[n] = (uint64_t)llv | ((uint64_t)mlv << 17) | ((uint64_t)offset << 34);
    }
#undef ZP_PEEK
#undef ZP_NORM
    if (bad & 2) return ZE_PARAM_UNSUPPORTED;           // an offset does not fit the packed form (window > 1 GiB)
    if (bad & 1) return ZE_CORRUPTION;
    if (left != 0) return ZE_CORRUPTION;
    *pNbSeq = nbSeq;
    return 0;
}

ZH_DEVFN void zp_seq_body(const ZhipPipeArgs& a, uint8_t* ldsBase, uint32_t* llBase, uint32_t* mlBase, uint8_t* llBits, uint8_t* mlBits)
{
    const uint32_t lane = zh_lane();
    if (lane < 36) { llBase[lane] = zc_llBase[lane]; llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { mlBase[lane] = zc_mlBase[lane]; mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    if (lane >= ZP_K2_LANES) return;
    ZpLaneLDS* Ll = (ZpLaneLDS*)(ldsBase + (size_t)lane * ZP_K2_LANE_LDS);
    for (;;) {                                                            // every lane steals its own frames
        const uint32_t i = zh_atomic_add(a.counters + 1, 1u);
        if (i >= a.count) break;
        ZdMeta* m = a.meta + i;
        if (m->path != 1) continue;
        const uint32_t f = a.first + i;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        uint32_t nbSeq = 0;
        const int err = zp_decode_sequences(src + m->seqOff, src + m->seqEnd, Ll, llBase, mlBase, llBits, mlBits,
                                            a.seqArena + (size_t)i * ZP_SEQ_CAP, &nbSeq);
        if (err == ZE_PARAM_UNSUPPORTED) {                                 // let the generic kernel handle it
            m->path = 2;
            const uint32_t k = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[k] = f;
        } else if (err) { m->status = err; m->path = 0; a.status[f] = err; a.outSizes[f] = 0; }
        else m->nbSeq = nbSeq;
    }
}

// ------------------------------------------------------------------------------------------ K3 (one wave per frame)
struct ZpExecLDS { uint8_t asmb[ZD_ASM_BYTES + 64]; uint16_t mBeg[64]; uint16_t mEnd[64]; uint32_t misc[8]; };

ZH_DEVFN int zp_exec_frame(const ZhipPipeArgs& a, ZpExecLDS& L, uint32_t i, uint32_t* pProduced)
{
    const uint32_t lane = zh_lane();
    const ZdMeta m = a.meta[i];
    const uint32_t f = a.first + i;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
    const uint64_t* seqs = a.seqArena + (size_t)i * ZP_SEQ_CAP;
    const bool litRLE = m.litMode == 2;
    const uint32_t rleByte = m.litOff;
    const uint8_t* litPtr = m.litMode == 0 ? src + m.litOff : a.litArena + (size_t)i * ZP_LIT_STRIDE;
    const uint8_t* dictEnd = dst;                         // no dictionary on this path
    uint8_t* const asmb = L.asmb;
    uint32_t op = 0, lp = 0, done = 0;
    const uint32_t nbSeq = m.nbSeq;
    while (done < nbSeq) {
        const uint32_t avail = nbSeq - done < 64 ? nbSeq - done : 64;
        uint32_t myLL = 0, myML = 0, myOF = 1;
        if (lane < avail) { const uint64_t q = seqs[done + lane]; myLL = (uint32_t)q & 0x1FFFF; myML = (uint32_t)(q >> 17) & 0x1FFFF; myOF = (uint32_t)(q >> 34); }
        uint32_t incL = zh_scan_add(myLL), incT = zh_scan_add(myLL + myML);
        // how many of these fit the assembly buffer
        const uint64_t fits = zh_ballot(lane < avail && incT <= ZD_ASM_BYTES);
        uint32_t cnt = (uint32_t)zh_popc64(fits);          // fits is a prefix mask (incT is monotone)
        const bool big = cnt == 0;
        if (big) cnt = 1;
        const bool act = lane < cnt;
        if (!act) { myLL = 0; myML = 0; myOF = 1; }
        const uint32_t totL = zh_shfl(incL, cnt - 1), totT = zh_shfl(incT, cnt - 1);
        if (lp + totL > m.litSize) return ZE_CORRUPTION;
        if ((uint64_t)op + totT > cap) return ZE_DST_TOO_SMALL;
        if (op + totT > m.blockMax) return ZE_CORRUPTION;
        const uint32_t litStart = lp + incL - myLL;
        const uint32_t oRel = incT - (myLL + myML), mRel = oRel + myLL;
        if (zh_ballot(act && (uint64_t)myOF > (uint64_t)op + mRel)) return ZE_CORRUPTION;
        if (big) {
            const uint32_t bll = zh_shfl(myLL, 0), bml = zh_shfl(myML, 0), bof = zh_shfl(myOF, 0);
            if (litRLE) zd_fill_wave(dst + op, rleByte, bll); else zd_copy_wave(dst + op, litPtr + lp, bll);
            zd_fence();
            zd_match_wave(dst, dictEnd, op + bll, bof, bml);
            zd_fence();
            op += totT; lp += totL; done += 1;
            continue;
        }
        const int32_t sAbs = (int32_t)(op + mRel) - (int32_t)myOF;
        const bool hasM = act && myML > 0;
        const bool farM = hasM && sAbs + (int32_t)myML <= (int32_t)op;
        {
            uint64_t rl[4], rm[4];
            const bool shortL = act && myLL > 0 && myLL <= ZD_COOP_LEN;
            const bool shortFar = farM && myML <= ZD_COOP_LEN;
            if (shortL && !litRLE) zd_ld32(litPtr + litStart, myLL, rl);
            if (shortFar) zd_ld32(dst + sAbs, myML, rm);
            if (shortL) {
                if (litRLE) { for (int k = 0; k < 4; k++) rl[k] = 0x0101010101010101ull * rleByte; }
                zd_st32(asmb + oRel, myLL, rl);
            }
            if (shortFar) zd_st32(asmb + mRel, myML, rm);
        }
        for (uint64_t mk = zh_ballot(act && myLL > ZD_COOP_LEN); mk; mk &= mk - 1) {
            const uint32_t l = (uint32_t)zh_ctz64(mk);
            const uint32_t d = zh_shfl(oRel, l), s = zh_shfl(litStart, l), n = zh_shfl(myLL, l);
            if (litRLE) { for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = (uint8_t)rleByte; }
            else { for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = litPtr[s + j]; }
        }
        for (uint64_t mk = zh_ballot(farM && myML > ZD_COOP_LEN); mk; mk &= mk - 1) {
            const uint32_t l = (uint32_t)zh_ctz64(mk);
            const uint32_t d = zh_shfl(mRel, l), n = zh_shfl(myML, l);
            const uint32_t s = zh_shfl((uint32_t)sAbs, l);
            for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = dst[s + j];
        }
        // ---- matches that read this batch's own output. A match may start as soon as every near match whose output
        // it reads is done: `need` = the set of those sequences (contiguous index range found by binary search over the
        // batch-relative match extents), so the number of rounds is the dependency depth, not the batch length.
        L.mBeg[lane] = (uint16_t)(act ? mRel : 0xFFFF); L.mEnd[lane] = (uint16_t)(act ? mRel + myML : 0xFFFF);
        zh_sync();
        bool pending = hasM && !farM;
        uint64_t need = 0;
        if (pending) {
            const uint32_t a0 = sAbs > (int32_t)op ? (uint32_t)(sAbs - (int32_t)op) : 0;          // first batch byte I read
            uint32_t b0 = (uint32_t)(sAbs + (int32_t)myML - (int32_t)op);                          // one past the last byte I read
            if (b0 > mRel) b0 = mRel;                                                              // my own output is handled by me
            uint32_t lo = 0, hi = 0;            // lo = first j with mEnd[j] > a0 ; hi = first j with mBeg[j] >= b0
            for (uint32_t stp = 32; stp; stp >>= 1) { if (lo + stp <= 64 && L.mEnd[lo + stp - 1] <= a0) lo += stp; }
            for (uint32_t stp = 32; stp; stp >>= 1) { if (hi + stp <= 64 && L.mBeg[hi + stp - 1] < b0) hi += stp; }
            if (hi > lane) hi = lane;           // only earlier sequences can feed me
            if (lo < hi) need = (hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & ~((1ull << lo) - 1);
        }
        const uint64_t nearMask = zh_ballot(pending);
        need &= nearMask;                       // literals and far matches are already in the buffer
        uint64_t doneMask = ~nearMask;
        for (;;) {
            const uint64_t pend = zh_ballot(pending);
            if (!pend) break;
            const uint64_t longReady = zh_ballot(pending && myML > ZD_COOP_LEN && (need & ~doneMask) == 0);
            if (longReady) {
                // whole wave copies one long ready match
                const uint32_t pf = (uint32_t)zh_ctz64(longReady);
                const uint32_t Frel = zh_shfl(mRel, pf), fml = zh_shfl(myML, pf), fof = zh_shfl(myOF, pf);
                const int32_t fs = (int32_t)(op + Frel) - (int32_t)fof;
                if (fof >= 64) {
                    for (uint32_t c = 0; c < fml; c += 64) {
                        const uint32_t j = c + lane;
                        if (j < fml) { const int32_t sp = fs + (int32_t)j; asmb[Frel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : dst[sp]; }
                        if (fof < fml) zh_sync();
                    }
                } else {
                    uint32_t idx = lane % fof; const uint32_t adv = 64 % fof;
                    for (uint32_t j = lane; j < fml; j += 64) {
                        const int32_t sp = fs + (int32_t)idx;
                        asmb[Frel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : dst[sp];
                        idx += adv; if (idx >= fof) idx -= fof;
                    }
                }
                if (lane == pf) pending = false;
                doneMask |= 1ull << pf;
            } else {
                const bool ready = pending && myML <= ZD_COOP_LEN && (need & ~doneMask) == 0;
                if (ready) {
                    if (sAbs >= (int32_t)op && myOF >= myML) {
                        uint64_t rr[4];
                        zd_ld32(asmb + (sAbs - (int32_t)op), myML, rr);
                        zd_st32(asmb + mRel, myML, rr);
                    } else {
                        for (uint32_t j = 0; j < myML; j++) {
                            const int32_t sp = sAbs + (int32_t)j;
                            asmb[mRel + j] = sp >= (int32_t)op ? asmb[sp - (int32_t)op] : dst[sp];
                        }
                    }
                    pending = false;
                }
                doneMask |= zh_ballot(ready);
            }
            zh_sync();
        }
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
        zh_sync();
        op += totT; lp += totL; done += cnt;
    }
    const uint32_t rest = m.litSize - lp;
    if ((uint64_t)op + rest > cap) return ZE_DST_TOO_SMALL;
    if (op + rest > m.blockMax) return ZE_CORRUPTION;
    if (litRLE) zd_fill_wave(dst + op, rleByte, rest); else zd_copy_wave(dst + op, litPtr + lp, rest);
    op += rest;
    const uint64_t fcs = (uint64_t)m.fcsLo | ((uint64_t)m.fcsHi << 32);
    if (fcs != ~0ull && fcs != op) return ZE_CORRUPTION;
    if (m.hasChecksum) {
        zd_fence();
        zh_sync();
        if (zh_opaque(lane) == 0) L.misc[0] = (uint32_t)ze_xxh64(dst, op);
        zh_sync();
        const uint32_t digest = zh_first(L.misc[0]);
        zh_sync();
        if (digest != m.checksum) return ZE_CHECKSUM_WRONG;
    }
    *pProduced = op;
    return 0;
}

ZH_DEVFN void zp_exec_body(const ZhipPipeArgs& a, ZpExecLDS& L)
{
    const uint32_t lane = zh_lane();
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counters + 2, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        const uint32_t i = zh_first(L.misc[7]);
        zh_sync();
        if (i >= a.count) break;
        if (a.meta[i].path != 1) continue;
        uint32_t produced = 0;
        const int err = zp_exec_frame(a, L, i, &produced);
        zh_sync();
        if (zh_opaque(lane) == 0) { a.status[a.first + i] = err; a.outSizes[a.first + i] = err ? 0 : produced; }
    }
}
