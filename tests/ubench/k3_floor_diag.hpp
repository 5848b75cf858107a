// DIAGNOSTIC ONLY (wrong output bytes; bench.py runs it under ZHIP_BENCH_NO_VERIFY) -- round 5, VERDICT r04 item 1's pre-flight question:
// how fast could K3 be if its instruction stream were (nearly) free? This is zp_exec_block with everything removed that is not memory traffic:
// per batch it loads the sequences, runs the two scans that place every item, issues the SAME global loads the real kernel issues -- the own-lane
// pieces of literal runs and far matches / pre-batch parts of up to 16 bytes, one 16-byte load per 64-byte line of every longer item -- folds
// them into one register, and writes the batch's output range with the same coalesced 16-byte stores (garbage: nothing is assembled). No LDS
// assembly, no unit dealing, no need-masks, no dependency rounds: ~150 instructions per batch instead of ~700. Its time is what the memory
// system (452 M far-match gathers per 65 536-frame step behind the L2) allows with K3's occupancy and access pattern: the floor under ANY
// rewrite of the batch body, hand-written ISA included. Built with -DZP_K3_DIAG_FLOOR (csrc/build_variants.sh floor).
// -DZP_FLOOR_WIN=<bytes> models an LDS WINDOW over the frame's recent output (the assembly buffer as a sliding linear buffer of that size: batches
// are appended, and when the next one might not fit the last ZP_FLOOR_KEEP bytes move to the front): an item whose source starts inside the window
// is read from LDS instead of global memory -- here: its global loads are dropped and two LDS reads stand in for them. The occupancy such a buffer
// costs is set with -DZP_FLOOR_LDSPAD=<bytes> (ZpExecLDS grows by it) and -DZP_K3_MINWAVES.
#ifndef ZP_FLOOR_WIN
#define ZP_FLOOR_WIN 0
#endif
#ifndef ZP_FLOOR_KEEP
#define ZP_FLOOR_KEEP 4096
#endif
template <bool DICT, bool PROF, bool MB>
ZH_DEVFN int zp_exec_block_floor(const ZhipPipeArgs& a, ZpExecLDS& L, const ZdMeta& m, uint32_t t, const uint8_t* src, uint8_t* dst, uint32_t cap, uint64_t cap64,
                                 uint32_t blockMax, uint32_t& opRef)
{
    const uint32_t lane = zh_lane();
    const uint64_t* seqs = a.seqArena + a.bases[2 * (size_t)t];
    const uint8_t* litPtr = m.litMode == 0 ? src + m.litOff : a.litArena + (size_t)a.bases[2 * (size_t)t + 1] * 16;
    const uint32_t litSize = m.litSize;
    uint32_t op = 0, lp = 0, done = 0, carry = 0;
    uint32_t base = 0;                                    // (window model) bytes of earlier output in the buffer in front of the batch
    const uint32_t nbSeq = m.nbSeq;
    uint64_t acc = 0;
    uint64_t qNext = lane < nbSeq ? seqs[lane] : 0;
    while (done < nbSeq) {
        const uint32_t avail = nbSeq - done < 64 ? nbSeq - done : 64;
        uint32_t myLL = 0, myML = 0, myOF = 1;
        if (lane < avail) { const uint64_t q = qNext; myLL = ZP_SEQ_LL(q); myML = ZP_SEQ_ML(q); myOF = ZP_SEQ_OF(q); }
        const uint32_t incL = zh_scan_add(myLL), incT = zh_scan_add(myLL + myML);
        const uint64_t fits = zh_ballot(lane < avail && incT + carry <= ZP_ASM_BYTES);
        uint32_t cnt = (uint32_t)zh_popc64(fits);
        if (cnt == 0) cnt = 1;
        qNext = done + cnt + lane < nbSeq ? seqs[done + cnt + lane] : 0;
        const bool act = lane < cnt;
        if (!act) { myLL = 0; myML = 0; myOF = 1; }
        const uint32_t totL = zh_bcast(incL, cnt - 1), totT = zh_bcast(incT, cnt - 1);
        if ((uint64_t)op + totT > cap || lp + totL > litSize) return ZE_CORRUPTION;
        const uint32_t litStart = lp + incL - myLL;
        const uint32_t ob = op - carry;
        const uint32_t mRel = incT - myML + carry;
        int32_t sAbs = (int32_t)(ob + mRel) - (int32_t)myOF;
        if (sAbs < 0) sAbs = 0;
        const bool hasM = act && myML > 0;
        const bool farM = hasM && sAbs + (int32_t)myML <= (int32_t)ob;
        const bool pre = hasM && !farM && sAbs < (int32_t)ob;
        uint32_t lenMi = farM ? myML : pre ? (uint32_t)((int32_t)ob - sAbs) : 0u;
        if (ZP_FLOOR_WIN) {                                // items that start inside the window: LDS reads instead of global loads
            const bool inWin = lenMi > 0 && sAbs >= (int32_t)ob - (int32_t)base;
#ifndef ZP_FLOOR_FREE          // (-DZP_FLOOR_FREE: the window's own LDS work left out -- the pure effect of the dropped requests)
            const uint8_t* w = L.asmb + (inWin ? (uint32_t)(sAbs & 1023) : 0u);
            acc ^= zh_ld64(w) ^ zh_ld64(w + (inWin && lenMi >= 8 && lenMi <= 16 ? lenMi - 8 : 0u));
#endif
            if (inWin) lenMi = 0;
        }
        // own-lane pieces (first + last 8 bytes), as the real kernel addresses them
        const bool shortL = act && myLL > 0 && myLL <= 16, shortM = lenMi > 0 && lenMi <= 16;
        const uint8_t* ql = litPtr + (shortL ? litStart : 0u);
        const uint8_t* qm = (shortM && (uint64_t)sAbs + 32 <= cap64) ? dst + sAbs : dst;
        const uint64_t l0 = zh_ld64(ql), l1 = zh_ld64(ql + (shortL && myLL >= 8 ? myLL - 8 : 0u));
        const uint64_t m0 = zh_ld64(qm), m1 = zh_ld64(qm + (shortM && lenMi >= 8 ? lenMi - 8 : 0u));
        acc ^= l0 ^ l1 ^ m0 ^ m1;
        // longer items: one 16-byte load per 64 bytes (the lines the unit pass would fetch)
        uint32_t lenL = act && myLL > 16 ? myLL : 0u, lenM = lenMi > 16 ? lenMi : 0u;
        if ((uint64_t)sAbs + lenM + 16 > cap64) lenM = 0;
        if (litStart + lenL + 16 > litSize + 256) lenL = 0;
        for (uint32_t k = 0; zh_ballot(k < lenL || k < lenM); k += 64) {
            if (k < lenL) { const zh_v16 v = zh_ld128(litPtr + litStart + k); acc ^= v.lo ^ v.hi; }
            if (k < lenM) { const zh_v16 v = zh_ld128(dst + sAbs + k); acc ^= v.lo ^ v.hi; }
        }
        *(uint64_t*)(L.asmb + 16 * lane) = acc; *(uint64_t*)(L.asmb + 16 * lane + 8) = acc;
        zh_sync();
        qNext = zh_opaque64(qNext);
        const uint32_t totB = totT + carry, whole = totB & ~15u;
        {
            uint8_t* out = dst + ob;
            for (uint32_t j = lane * 16; j < whole; j += 1024) { const zh_v16 v = *(const zh_v16*)(L.asmb + (j & 1023u)); *(zh_v16*)(out + j) = v; }
        }
        carry = totB - whole;
        zh_sync();
        if (ZP_FLOOR_WIN) {
            base += whole;
            if (base + ZP_ASM_BYTES + 64 > ZP_FLOOR_WIN) {                 // the buffer is full: the last KEEP bytes move to the front
                if (base > ZP_FLOOR_KEEP) base = ZP_FLOOR_KEEP;
#ifndef ZP_FLOOR_FREE
                zh_v16 r[ZP_FLOOR_KEEP / 1024];
                for (uint32_t k = 0; k < ZP_FLOOR_KEEP / 1024; k++) r[k] = *(const zh_v16*)(L.asmb + (((k + 1) * 1024 + lane * 16) & 4095u));
                zh_sync();
                for (uint32_t k = 0; k < ZP_FLOOR_KEEP / 1024; k++) *(zh_v16*)(L.asmb + ((k * 1024 + lane * 16) & 4095u)) = r[k];
                zh_sync();
#endif
            }
        }
        op += totT; lp += totL; done += cnt;
    }
    const uint32_t rest = litSize - lp;
    if ((uint64_t)op + rest > cap) return ZE_DST_TOO_SMALL;
    zd_copy_wave(dst + op, litPtr + lp, rest);
    op += rest;
    if (acc == 0x123456789ABCDEFull) dst[0] = 1;          // (keeps the loads alive)
    opRef = op;
    return 0;
}
