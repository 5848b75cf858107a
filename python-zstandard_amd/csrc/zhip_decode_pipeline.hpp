// zhip_decode_pipeline.hpp -- phase-split batch decoder: frames made of ONE block (every frame multi_compress_to_buffer produces for
// inputs <= 128 KiB) in the kernels listed below, frames of SEVERAL blocks in their several-block instantiations (the work item of K1b / K2
// is a block; zhip_format.hpp, ZpFrameRec), with or without a dictionary. Profile-driven (profiles/r01a): the fused one-wave-per-frame kernel spends 63 % of its
// cycles in the single-lane tANS chain and is limited to ~8-11 frames per CU by its 14 KiB of LDS. Splitting the frame loop by phase lets
// each phase use the lane mapping -- and the LDS / register budget -- that fits ITS serial chain (DESIGN.md 4.1 has the measurements):
//
//   K1  zhip_decode_lit_kernel   one wave per frame   : frame / block / literals headers; Huffman weights -> decoding table (wave-parallel) into
//                                                       the frame's slot of the table arena; sequences header -> three FSE tables (wave-parallel)
//                                                       into its 2.5 KiB slot; raw / RLE single-block frames finish here. Frames of a dictionary
//                                                       batch whose tables are all "repeat" / "treeless" only get a flag: the tables are the dictionary's
//   KB  zhip_decode_bin_kernel   128 waves per chunk  : work orders -- frames by decreasing sequence count (K2's) and literal count (K1b's), from
//                                                       the bin counters K1 filled, so lanes sharing a wave run equally long
//   K1b zhip_decode_huf_kernel   4 LANES per frame    : the (up to) four Huffman streams, 8 frames per wave, tables in LDS (3 KiB each), 6 waves per CU
//   K2  zhip_decode_seq_kernel   a QUAD per frame     : the serial tANS chain with its three streams side by side (lanes OF / ML / LL / spare), 15
//                                                       frames per wave, four waves per CU: the 60 frames' tables ARE the CU's 160 KiB of LDS.
//                                                       Emits packed 8-byte sequences (zp_seqq_body)
//   K3  zhip_decode_exec_kernel  one wave per frame   : 64 sequences per batch, the batch's output assembled in LDS and flushed in whole 16-byte
//                                                       units; 77 VGPRs, six waves per SIMD (zhip_decode_exec_dict_kernel: with a dictionary)
//
// Several-block mode (the caller's size hint exceeds one block): zhip_decode_lit_mb_kernel (K1, a wave per frame over its blocks),
// zhip_decode_seq_mb_kernel (K2 per block, symbolic repeat-offset history), zhip_decode_exec_mb_kernel (K3, a wave per frame over its blocks).
// Literals (K1b / K1 -> K3) and packed sequences (K2 -> K3) live in ONE compact arena per chunk (round 5; ZhipPipeArgs.bases): K1 claims a frame's
// literal room from its end, K2 a 15-frame group's sequence room from its start, one atomic add each.
// Anything else (sources of 2 GiB and more, frames that find the chunk's block slots or the arena's room used up, offsets the packed form cannot hold
// in a frame of several blocks, frames of several blocks without the size hint) goes to the generic fused kernel through a fallback list, so
// results are identical on every input.
#pragma once
#include "zhip_decode_kernel.hpp"

// LDS cells of the table builder (base | nbBits << 10 | ... | symbol << 19) -> K2's 2-byte cells (symbol << 10 | x), x = (base + size) >> nbBits, two per store
ZH_DEV void zp_pack_fse(ZdLDS& L, uint32_t* T, uint32_t llLog, uint32_t ofLog, uint32_t mlLog)
{
    const uint32_t lane = zh_lane();
    for (uint32_t u = lane; u < ZP_FSE_CELLS / 2; u += 64) {
        const uint32_t lg = u < 256 ? llLog : u < 512 ? mlLog : ofLog;
        const uint32_t local = (2 * u) & (u < 512 ? 511u : 255u);
        uint32_t pair = 0;
        for (uint32_t k = 0; k < 2; k++) {
            uint32_t cell = 0;
            if (local + k < (1u << lg)) {
                const uint32_t e = L.fse[2 * u + k];
                cell = ((e >> 19) << 10) | (((e & 1023) + (1u << lg)) >> ((e >> 10) & 15));
            }
            pair |= cell << (16 * k);
        }
        T[u] = pair;
    }
}

// ------------------------------------------------------------------------------------------ K1
// frame header (RFC 8878 3.1.1.1; ZSTD_getFrameHeader_advanced zstd.c:43682, the checks of ZSTD_decompressFrame :44174): where the first
// block header lies, the block maximum, the content size (~0 = not in the header), the checksum flag. 0 or a zstd error code.
struct ZpHdr { uint32_t pos, blockMax, hasChecksum, skippable; uint64_t fcs; };
ZH_DEV int zp_frame_header(const ZhipPipeArgs& a, const uint8_t* src, uint32_t srcSize, ZpHdr& h)
{
    const uint32_t mg = a.magicless ? 0u : 4u;                     // ZSTD_f_zstd1_magicless: the frame starts at its descriptor byte
    h.skippable = 0;
    if (srcSize < mg + 1) return ZE_SRC_SIZE_WRONG;
    if (mg && (zh_ld32(src) & 0xFFFFFFF0u) == ZF_MAGIC_SKIPPABLE) {  // a skippable frame: ZSTD_decompressStream passes over it and stops at the frame
        if (srcSize < 8 || (uint64_t)zh_ld32(src + 4) + 8 > srcSize) return ZE_SRC_SIZE_WRONG;     // boundary -- nothing produced (zstd.c:43706, :44731)
        h.skippable = 1; h.pos = 0; h.blockMax = 0; h.hasChecksum = 0; h.fcs = 0;
        return 0;
    }
    if (mg && zh_ld32(src) != ZF_MAGIC) return ZE_PREFIX_UNKNOWN;
    const uint32_t fhd = src[mg];
    const uint32_t dictCode = fhd & 3, single = (fhd >> 5) & 1, fcsCode = fhd >> 6;
    h.hasChecksum = (fhd >> 2) & 1;
    const uint32_t dictBytes = dictCode == 3 ? 4 : dictCode;
    const uint32_t fcsBytes = fcsCode == 0 ? single : (1u << fcsCode);
    const uint32_t hs = mg + 1 + (single ? 0 : 1) + dictBytes + fcsBytes;
    if (fhd & 8) return ZE_FRAMEPARAM_UNSUPPORTED;
    if (srcSize < hs) return ZE_SRC_SIZE_WRONG;
    uint32_t pos = mg + 1;
    uint64_t windowSize = 0;
    if (!single) {
        const uint32_t wd = src[pos++], wl = 10 + (wd >> 3);
        if (wl > 31) return ZE_WINDOW_TOO_LARGE;
        windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (wd & 7);
    }
    {   const uint32_t dictID = dictCode == 0 ? 0u : dictCode == 1 ? src[pos] : dictCode == 2 ? zh_ld16(src + pos) : zh_ld32(src + pos);
        pos += dictBytes;
        if (dictID && dictID != a.dictID) return ZE_DICT_WRONG; }     // ZSTD_decompressFrame's check (zstd.c:44246)
    uint64_t fcs = ~0ull;
    if (fcsCode == 0) { if (single) fcs = src[pos]; }
    else if (fcsCode == 1) fcs = (uint64_t)zh_ld16(src + pos) + 256;
    else if (fcsCode == 2) fcs = zh_ld32(src + pos);
    else fcs = zh_ld64(src + pos);
    pos += fcsBytes;
    if (single) windowSize = fcs;
    if (windowSize > a.maxWindowSize) return ZE_WINDOW_TOO_LARGE;
    h.blockMax = windowSize < ZF_BLOCK_MAX ? (uint32_t)windowSize : ZF_BLOCK_MAX;
    h.fcs = fcs; h.pos = pos;
    return 0;
}

// One compressed block, content [pos, pos + bs) of the frame at `src`, for the later kernels: the literals section -> Huffman table into
// `df.table` (or literals that need no K1b), the sequences header -> the three FSE tables into slot `t` of the table arena; fills the
// item's record `m` (offsets relative to `src`). `st` carries what a block inherits (the Huffman weights' count, the table logs; the
// tables themselves lie in L). `share`: a dictionary frame's "repeat" / "treeless" tables may be read where the dictionary's lie.
// `dictInLds`: the dictionary's FSE tables are already in L.fse (several-block frames load them once). All lanes call; 0 or an error code.
ZH_DEVFN int zp_block_tables(const ZhipPipeArgs& a, ZdLDS& L, ZdState& st, ZdLitDefer& df, const uint8_t* src, uint32_t pos, uint32_t bs,
                             uint32_t blockMax, uint32_t t, ZdMeta& m, ZdProf& P, bool share, bool dictInLds)
{
    const uint32_t lane = zh_lane();
    uint8_t* lit;
    {
        // compact arena (shared with K2's sequence rooms): the section header says how many literals a compressed / treeless section regenerates (RFC 8878 3.1.1.3.1.1; raw
        // and RLE sections need no room: they are read in place / filled). + 256 bytes: K1b's whole-unit stores and K3's over-reads stay inside
        uint32_t need16 = 0;
        if (bs >= 5 && (src[pos] & 3u) >= 2u) {
            const uint32_t v = zh_ld32(src + pos), fmt = (v >> 2) & 3u;
            const uint32_t regen = fmt < 2 ? (v >> 4) & 0x3FFu : fmt == 2 ? (v >> 4) & 0x3FFFu : (v >> 4) & 0x3FFFFu;
            if (regen <= ZF_BLOCK_MAX) need16 = (regen + 256u + 15u) >> 4;                     // (larger: zd_literals refuses the block below)
        }
        uint32_t lb = 0;
        if (need16) {
            lb = zh_first(zh_atomic_add(a.counters + 8, lane == 0 ? need16 : 0u));
            if ((uint64_t)lb + need16 > a.arenaBudget16) {                                     // the chunk's room is used up: the generic kernel's frame
                // (the refused claim is taken back: K2's room check adds this counter to its own, and a claim that got nothing must not shut every K2 group of the chunk out -- ADVICE r05)
                if (zh_opaque(lane) == 0) zh_atomic_add(a.counters + 8, 0u - need16);
                return ZP_RC_FALLBACK;
            }
        }
        // (literal rooms grow DOWN from the arena's end, K2's sequence rooms UP from its start: one budget, and each kind stays packed -- K2's stores
        // are what feels a wider destination, section 4.1)
        const uint32_t lpos = need16 ? a.arenaBudget16 - lb - need16 : 0u;
        lit = a.litArena + (size_t)lpos * 16;
        if (zh_opaque(lane) == 0) a.bases[2 * (size_t)t + 1] = lpos;
    }
    st.litPtr = lit; st.litSize = 0; st.litRLE = 0; st.rleByte = 0;
    const ZhipDictEntropy* const de = a.dictEntropy;
    const int r = zd_literals(L, st, src + pos, bs, lit, blockMax, P, &df);
    if (r < 0) return -r;
    m.litSize = st.litSize;
    if (df.taken) { m.litMode = 3u | (df.log << 8) | (df.four << 16) | (df.shared ? ZP_LIT_SHARED : 0u); m.litOff = (uint32_t)(df.streams - src); m.produced = df.streamBytes; }
    else if (st.litRLE) { m.litMode = 2; m.litOff = st.rleByte; }
    else if (st.litPtr == lit) { m.litMode = 1; m.litOff = 0; }
    else { m.litMode = 0; m.litOff = (uint32_t)(st.litPtr - src); }
    // sequences header (RFC 8878 3.1.1.3.2): count, compression modes, table descriptions -> FSE tables for K2
    const uint8_t* sp = src + pos + (uint32_t)r; const uint8_t* const send = src + pos + bs;
    if (sp >= send) return ZE_SRC_SIZE_WRONG;
    uint32_t nbSeq = *sp++;
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (sp + 2 > send) return ZE_SRC_SIZE_WRONG; nbSeq = zh_ld16(sp) + 0x7F00; sp += 2; }
        else { if (sp >= send) return ZE_SRC_SIZE_WRONG; nbSeq = ((nbSeq - 128) << 8) + *sp++; }
    }
    m.nbSeq = nbSeq;
    if (nbSeq == 0) { if (sp != send) return ZE_CORRUPTION; }
    else {
        if (nbSeq > ZP_SEQ_CAP - 16) return ZE_CORRUPTION;      // (> 43 690 cannot fit a block; K2 stores a few slots past the longest frame of its wave and parks idle lanes' stores in the last)
        if (sp >= send) return ZE_SRC_SIZE_WRONG;
        const uint32_t modes = *sp++;
        if (modes & 3) return ZE_CORRUPTION;
        // every table "repeat" in a dictionary frame: the tables ARE the dictionary's, ready-made in K2's form (ZhipDictTables.fseK2):
        // nothing to build, nothing to write -- K2 copies them from there
        const bool allShared = share && de && (modes >> 2) == 0x3F;
        if (allShared) {
            st.llLog = de->llLog; st.ofLog = de->ofLog; st.mlLog = de->mlLog;
            if (sp >= send) return ZE_CORRUPTION;
        } else {
            if (de && !dictInLds) {                                 // "repeat" takes the dictionary's table: drop it into its LDS place
                zh_sync();
                const uint32_t* T = a.dictTables->fse;
                if ((modes >> 6) == 3) { for (uint32_t k = lane; k < (1u << de->llLog); k += 64) L.fse[ZD_FSE_LL + k] = T[ZD_FSE_LL + k]; st.llLog = de->llLog; }
                if (((modes >> 4) & 3) == 3) { for (uint32_t k = lane; k < (1u << de->ofLog); k += 64) L.fse[ZD_FSE_OF + k] = T[ZD_FSE_OF + k]; st.ofLog = de->ofLog; }
                if (((modes >> 2) & 3) == 3) { for (uint32_t k = lane; k < (1u << de->mlLog); k += 64) L.fse[ZD_FSE_ML + k] = T[ZD_FSE_ML + k]; st.mlLog = de->mlLog; }
                zh_sync();
            }
            // (round 4: the three table descriptions -- at most ~150 bytes together, the format bounds them -- are parsed from an LDS copy: in
            // place every field was a dependent global-memory round trip of lane 0, zd_read_huf_weights has the arithmetic. The upper half of the
            // union is free here: the Huffman table has left for the table arena, zd_build_fse uses symAt in the lower half)
            const uint32_t hn = (uint32_t)(send - sp) < 256u ? (uint32_t)(send - sp) : 256u;
            uint8_t* const stg = (uint8_t*)L.u.huf + 4096;
            zh_sync();
            zd_stage_wave(stg, sp, hn);
            zh_sync();
            const uint8_t* lp = stg; const uint8_t* const lend = stg + hn;
            // (K0's record, where one was made from THIS block's descriptions: the distributions come parsed, lane 0 walks nothing)
            const ZpPre* const spre = df.pre && df.pre->seqAt == (uint32_t)(sp - (src + pos)) ? df.pre : nullptr;
            int q = zd_seq_table(L, modes >> 6, ZD_KIND_LL, &st.llLog, lp, lend, spre); if (q < 0) return -q; sp += q; lp += q;
            q = zd_seq_table(L, (modes >> 4) & 3, ZD_KIND_OF, &st.ofLog, lp, lend, spre); if (q < 0) return -q; sp += q; lp += q;
            q = zd_seq_table(L, (modes >> 2) & 3, ZD_KIND_ML, &st.mlLog, lp, lend, spre); if (q < 0) return -q; sp += q; lp += q;
            if (sp >= send) return ZE_CORRUPTION;
            zh_sync();
            // LDS cells (base | nbBits << 10 | ...) -> 2-byte cells (symbol << 10 | x), x = (base + size) >> nbBits, two per store
            uint32_t* const T = (uint32_t*)(a.fseTables + (size_t)t * ZP_FSE_CELLS);
            zp_pack_fse(L, T, st.llLog, st.ofLog, st.mlLog);
        }
        m.logs = st.llLog | (st.ofLog << 8) | (st.mlLog << 16) | (allShared ? ZP_LOGS_SHARED : 0u);
    }
    m.seqOff = (uint32_t)(sp - src); m.seqEnd = pos + bs;
    ZD_T(P, ZP_SEQTAB);
    return 0;
}

// the item's place in K2's and K1b's work orders (KB below): its bin, and its rank inside the bin -- the atomic's return value (lane 0 calls)
ZH_DEV void zp_enter_bins(const ZhipPipeArgs& a, ZdMeta& m)
{
    const uint32_t ks = m.nbSeq ? 1u + (m.nbSeq >> ZP_BIN_SHIFT) : 0u;
    const uint32_t kl = (m.litMode & 255u) == 3u ? 1u + (m.litSize >> ZP_LITBIN_SHIFT) : 0u;
    if (ks) m.pad = zh_atomic_add(a.counters + ZP_CNT_BINS + (256 - (ks > 256 ? 256u : ks)), 1u);
    if (kl) m.hasChecksum |= zh_atomic_add(a.counters + ZP_CNT_BINS + 256 + (256 - (kl > 256 ? 256u : kl)), 1u) << 1;
}
// the same for a wave whose every lane holds a frame of its own (zp_lit_lanes_body; all lanes call, `on` = this lane has a record to enter): the lanes that enter the same
// bin claim their ranks with ONE atomic -- 262 144 four-KiB documents of one shape are two bins, and a per-lane atomic on one address is what the pass would then wait for
ZH_DEV void zp_enter_bins_wave(const ZhipPipeArgs& a, ZdMeta& m, bool on)
{
    const uint32_t lane = zh_lane();
    const uint32_t ks = on && m.nbSeq ? 1u + (m.nbSeq >> ZP_BIN_SHIFT) : 0u;
    const uint32_t kl = on && (m.litMode & 255u) == 3u ? 1u + (m.litSize >> ZP_LITBIN_SHIFT) : 0u;
    const uint32_t bS = ks ? 256u - (ks > 256u ? 256u : ks) : 0xFFFFu, bL = kl ? 256u + (256u - (kl > 256u ? 256u : kl)) : 0xFFFFu;
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const uint32_t mine = which ? bL : bS;
        uint64_t todo = zh_ballot(mine != 0xFFFFu);
        while (todo) {
            const uint32_t leader = (uint32_t)zh_ctz64(todo);
            const uint32_t b = zh_shfl(mine, leader);
            const uint64_t same = zh_ballot(mine == b);
            uint32_t base = 0;
            if (lane == leader) base = zh_atomic_add(a.counters + ZP_CNT_BINS + b, (uint32_t)zh_popc64(same));
            base = zh_shfl(base, leader);
            if (mine == b) { const uint32_t rank = base + (uint32_t)zh_popc64(same & zh_lt_mask()); if (which) m.hasChecksum |= rank << 1; else m.pad = rank; }
            todo &= ~same;
        }
    }
}
ZH_DEV void zp_meta_clear(ZdMeta& m)
{
    m.status = 0; m.path = 0; m.seqOff = m.seqEnd = 0; m.litSize = 0; m.litMode = 0; m.litOff = 0; m.nbSeq = 0;
    m.blockMax = 0; m.fcsLo = m.fcsHi = 0xFFFFFFFFu; m.produced = 0; m.hasChecksum = 0; m.checksum = 0; m.logs = 0; m.pad = 0;
}

// A LANE's walk over one frame of a dictionary batch (round 6: configs[3]'s 262 144 x 4 KiB documents spent 1.76 ms per launch in K1 with a whole wave per document,
// although nothing is built for them): a frame of ONE compressed block whose literals are raw, RLE or "treeless" on the dictionary's ready-made Huffman table and
// whose sequence tables are all "repeat" (the dictionary's, ZhipDictTables.fseK2) is nothing but header arithmetic -- frame header, block header, literals section
// header (RFC 8878 3.1.1.3.1.1), sequences header (3.1.1.3.2.1). Fills the record zp_lit_one would (same fields, same values) and says how much literal room the
// frame wants; ANYTHING else -- a table of its own, another block layout, any check that fails -- returns false and the wave takes the frame as before, so every
// error is found, and worded, by the code that always did. zstd.c:45767 (ZSTD_decodeLiteralsBlock), :46328 (ZSTD_decodeSeqHeaders).
ZH_DEV bool zp_lit_shared_try(const ZhipPipeArgs& a, uint32_t f, ZdMeta& m, uint32_t& need16)
{
    zp_meta_clear(m); need16 = 0;
    if (!a.dictEntropy->hufCount || a.dictTables->hufLog > ZP_HUF_LOGMAX) return false;       // (no ready-made Huffman table to share)
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    if (srcSize64 > 0x7FFFFFFFull) return false;
    const uint32_t srcSize = (uint32_t)srcSize64;
    ZpHdr h;
    if (zp_frame_header(a, src, srcSize, h) || h.skippable) return false;
    const uint32_t blockMax = h.blockMax;
    uint32_t pos = h.pos;
    m.blockMax = blockMax; m.fcsLo = (uint32_t)h.fcs; m.fcsHi = (uint32_t)(h.fcs >> 32);
    if (pos + 3 > srcSize) return false;
    const uint32_t bh = zh_ld24(src + pos); pos += 3;
    const uint32_t bs = bh >> 3;
    if (!(bh & 1) || ((bh >> 1) & 3) != 2) return false;
    if (pos + bs > srcSize || bs > blockMax || bs < 2) return false;
    const ZhipDictEntropy* const de = a.dictEntropy;
    const uint8_t* const b = src + pos;
    const uint32_t b0 = b[0], lt = b0 & 3, fmt = (b0 >> 2) & 3;
    uint32_t used, regen;
    if (lt == 2) return false;                                          // a Huffman table of its own
    if (lt < 2) {
        uint32_t hdr;
        if (fmt == 1) { hdr = 2; regen = zh_ld16(b) >> 4; }            // (bs >= 2)
        else if (fmt == 3) { if (bs < 3) return false; hdr = 3; regen = zh_ld24(b) >> 4; }
        else { hdr = 1; regen = b0 >> 3; }
        if (regen > blockMax) return false;
        if (lt == 0) { if (hdr + regen > bs) return false; m.litMode = 0; m.litOff = pos + hdr; used = hdr + regen; }
        else { if (hdr + 1 > bs) return false; m.litMode = 2; m.litOff = b[hdr]; used = hdr + 1; }
    } else {
        if (bs < 5) return false;
        const uint32_t v = zh_ld32(b);
        uint32_t hdr, four, csize;
        if (fmt < 2) { hdr = 3; four = fmt; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
        else if (fmt == 2) { hdr = 4; four = 1; regen = (v >> 4) & 0x3FFF; csize = v >> 18; }
        else { hdr = 5; four = 1; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) + ((uint32_t)b[4] << 10); }
        if (regen > blockMax || regen > ZF_BLOCK_MAX) return false;
        if (four ? regen < 6 : regen == 0) return false;
        if (hdr + csize > bs) return false;
        if (four && (csize < 10 || 3 * ((regen + 3) / 4) > regen)) return false;
        m.litMode = 3u | (a.dictTables->hufLog << 8) | (four << 16) | ZP_LIT_SHARED; m.litOff = pos + hdr; m.produced = csize;
        need16 = (regen + 256u + 15u) >> 4;
        used = hdr + csize;
    }
    m.litSize = regen;
    uint32_t sp = pos + used; const uint32_t send = pos + bs;
    if (sp >= send) return false;
    uint32_t nbSeq = src[sp++];
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (sp + 2 > send) return false; nbSeq = zh_ld16(src + sp) + 0x7F00; sp += 2; }
        else { if (sp >= send) return false; nbSeq = ((nbSeq - 128) << 8) + src[sp++]; }
    }
    m.nbSeq = nbSeq;
    if (nbSeq == 0) { if (sp != send) return false; }
    else {
        if (nbSeq > ZP_SEQ_CAP - 16 || sp >= send) return false;
        const uint32_t modes = src[sp++];
        if ((modes & 3) || (modes >> 2) != 0x3F) return false;          // a table of its own (or a reserved bit)
        if (sp >= send) return false;
        m.logs = de->llLog | (de->ofLog << 8) | (de->mlLog << 16) | ZP_LOGS_SHARED;
    }
    m.seqOff = sp; m.seqEnd = send;
    if (h.hasChecksum) {
        if (send + 4 > srcSize) return false;
        m.hasChecksum = 1; m.checksum = zh_ld32(src + send);
    }
    m.path = 1;
    return true;
}

// K1 of dictionary batches, first pass: a wave takes 64 frames, a lane walks each (zp_lit_shared_try); the frames a lane cannot finish are LISTED for K1 proper
// (zp_lit_body in list mode), which then runs only over them. A kernel of its own: folded into zp_lit_body the per-lane records pushed that kernel into 288 bytes
// of scratch per lane and made the wave-per-frame path 25 x slower (r06e: 2.2 -> 55 ms per 65 536 frames without any dictionary).
ZH_DEVFN void zp_lit_lanes_body(const ZhipPipeArgs& a)
{
    const uint32_t lane = zh_lane();
    bool anyCk = false;                                  // a frame this wave finished carries a content checksum: said ONCE, when the wave leaves (counters[12]; per-frame stores
                                                         // or atomics on that one word cost K1 0.5-3 ms per 65 536 frames, r06zs / r06zt)
    for (;;) {
        const uint32_t base = zh_first(zh_atomic_add(a.counters + 9, lane == 0 ? 64u : 0u));
        if (base >= a.count) { if (anyCk && lane == 0) *(volatile uint32_t*)(a.counters + 12) = 1u; break; }
        const uint32_t i = base + lane;
        const bool mine = i < a.count;
        ZdMeta m; uint32_t need16 = 0;
        const bool done = mine && zp_lit_shared_try(a, a.first + i, m, need16);
        // the frames' literal rooms: ONE claim for the 64 (zp_block_tables makes one per frame)
        const uint32_t n16 = done ? need16 : 0u;
        const uint32_t incl = zh_scan_add(n16), total = zh_shfl(incl, 63);
        uint32_t lb = 0;
        if (total) lb = zh_first(zh_atomic_add(a.counters + 8, lane == 0 ? total : 0u)) + incl - n16;
        const bool noRoom = done && n16 && (uint64_t)lb + n16 > a.arenaBudget16;    // the chunk's room is used up: the generic kernel's frame
        zp_enter_bins_wave(a, m, done && !noRoom);
        if (a.ckLater) {                                                            // (frames that carry a checksum: KX's count)
            const uint32_t nck = (uint32_t)zh_popc64(zh_ballot(done && !noRoom && (m.hasChecksum & 1u)));
            anyCk |= nck != 0;
        }
        if (done) {
            if (noRoom) {
                const uint32_t bm = m.blockMax, lo = m.fcsLo, hi = m.fcsHi;
                zp_meta_clear(m); m.blockMax = bm; m.fcsLo = lo; m.fcsHi = hi; m.path = 2;
                a.meta[i] = m;
                const uint32_t k = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[k] = a.first + i;
            } else {
                a.bases[2 * (size_t)i + 1] = n16 ? a.arenaBudget16 - lb - n16 : 0u;
                a.meta[i] = m;
#ifdef ZHIP_EMU
                zd_stat[7]++;                                               // (test hook [7]: frames a lane finished)
#endif
            }
        }
        // the others: one list claim for the wave
        const uint64_t rest = zh_ballot(mine && !done);
        if (rest) {
            const uint32_t at = zh_first(zh_atomic_add(a.counters + 10, lane == 0 ? (uint32_t)zh_popc64(rest) : 0u));
            if (mine && !done) a.order[at + (uint32_t)zh_popc64(rest & zh_lt_mask())] = i;
        }
    }
}

// ------------------------------------------------------------------------------------------ K0 (a LANE per frame: K1's serial parsers, 64 frames at a time)
// K1 gives a frame a whole wave, and 41 % of that wave's time was ONE lane walking bit fields: the Huffman weights' description (a distribution, a 64-cell FSE
// table, ~100 weights decoded by two interleaved states) and the three sequence distributions (`profiles/r06v_k1_fine_phase_timers.txt`). Those walks need no wave:
// here every lane does them for a frame of its own and leaves the results in the frame's ZpPre record; K1 then copies weights and counts and goes straight to its
// wave-parallel table builders. K0 checks what it needs to find the descriptions and to trust its own results, and on ANYTHING else -- a check that fails, a layout
// it does not know, a description that does not parse -- leaves that part of the record empty: K1 parses it itself and finds, and words, the error as it always
// did. Frames of one compressed block (the single-block pipeline's); RFC 8878 3.1.1.3.1 / 3.1.1.3.2 / 4.1.1 / 4.2.1, zstd.c:45767, :46328.
struct ZpPreLane { int16_t wnorm[64]; uint16_t cell[64]; uint8_t symAt[64]; uint8_t next[16]; uint32_t pad; };      // 340 bytes = 85 dwords: the lanes' arrays start in different banks
struct ZpPreLDS { ZpPreLane lane[64]; };

// the weights: distribution -> FSE decoding table (FSE_buildDTable's serial form: low-probability symbols from the top, the spread walk, states numbered in table order --
// cell for cell what zd_build_fse makes wave-parallel) -> the two-state decode of zd_read_huf_weights, over the description where it lies
ZH_DEV void zp_pre_weights(ZpPre& R, ZpPreLane& S, const uint8_t* desc, uint32_t hb, uint32_t at)
{
    uint32_t maxS = 63, tl = 0;
    const int r = zd_read_ncount_to(S.wnorm, desc, desc + hb, &maxS, &tl);
    if (r < 0 || tl > 6 || maxS > 12) return;
    const uint32_t size = 1u << tl, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t high = size - 1, total = 0;
    for (uint32_t q = 0; q <= maxS; q++) {
        const int n = S.wnorm[q];
        if (n == -1) { S.symAt[high--] = (uint8_t)q; S.next[q] = 1; total++; }
        else { S.next[q] = (uint8_t)n; total += (uint32_t)n; }
    }
    if (total != size) return;
    uint32_t pos = 0;
    for (uint32_t q = 0; q <= maxS; q++) {
        const int n = S.wnorm[q];
        for (int k = 0; k < n; k++) {
            S.symAt[pos] = (uint8_t)q;
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    for (uint32_t u = 0; u < size; u++) {
        const uint32_t q = S.symAt[u], x = S.next[q];
        S.next[q] = (uint8_t)(x + 1);
        const uint32_t nb = tl - (uint32_t)zh_highbit32(x);
        S.cell[u] = (uint16_t)(((x << nb) - size) | (nb << 6) | (q << 9));
    }
    ZdBits b; uint32_t cnt = 0;
    if (!zd_bits_init(b, desc + r, hb - (uint32_t)r)) return;
    zd_bits_reload(b);
    uint32_t s1 = zd_bits_peek(b, tl); b.used += tl;
    uint32_t s2 = zd_bits_peek(b, tl); b.used += tl;
    if (tl == 0) { s1 = s2 = 0; }
    int64_t left;
    for (;;) {
        if (b.used >= 32) zd_bits_reload(b);
        left = (int64_t)(b.ptr - b.start) * 8 + 64 - (int64_t)b.used;
        const uint32_t e1 = S.cell[s1], nb1 = (e1 >> 6) & 7;
        if (cnt > 253) return;
        R.weights[cnt++] = (uint8_t)(e1 >> 9);
        const uint32_t v1 = nb1 ? zd_bits_peek(b, nb1) : 0; b.used += nb1; left -= nb1;
        s1 = (e1 & 63) + v1;
        const uint32_t e2 = S.cell[s2], nb2 = (e2 >> 6) & 7;
        if (left < 0) { R.weights[cnt++] = (uint8_t)(e2 >> 9); break; }
        if (cnt > 253) return;
        R.weights[cnt++] = (uint8_t)(e2 >> 9);
        const uint32_t v2 = nb2 ? zd_bits_peek(b, nb2) : 0; b.used += nb2; left -= nb2;
        s2 = (e2 & 63) + v2;
        if (left < 0) { R.weights[cnt++] = (uint8_t)(S.cell[s1] >> 9); break; }
    }
    R.wAt = at; R.wCount = cnt;
}

ZH_DEV void zp_pre_one(const ZhipPipeArgs& a, uint32_t f, ZpPre& R, ZpPreLane& S)
{
    R.wCount = 0; R.wAt = 0; R.seqAt = 0; R.t[0].valid = 0; R.t[1].valid = 0; R.t[2].valid = 0;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
    if (srcSize64 > 0x7FFFFFFFull) return;
    const uint32_t srcSize = (uint32_t)srcSize64;
    ZpHdr h;
    if (zp_frame_header(a, src, srcSize, h) || h.skippable) return;
    uint32_t pos = h.pos;
    if (pos + 3 > srcSize) return;
    const uint32_t bh = zh_ld24(src + pos); pos += 3;
    const uint32_t bs = bh >> 3;
    if (!(bh & 1) || ((bh >> 1) & 3) != 2) return;                      // one block, compressed
    if (pos + bs > srcSize || bs > h.blockMax || bs < 2) return;
    const uint8_t* const b = src + pos;
    const uint32_t b0 = b[0], lt = b0 & 3, fmt = (b0 >> 2) & 3;
    uint32_t used, regen;
    if (lt < 2) {
        uint32_t hdr;
        if (fmt == 1) { hdr = 2; regen = zh_ld16(b) >> 4; }
        else if (fmt == 3) { if (bs < 3) return; hdr = 3; regen = zh_ld24(b) >> 4; }
        else { hdr = 1; regen = b0 >> 3; }
        if (regen > h.blockMax) return;
        if (lt == 0) { if (hdr + regen > bs) return; used = hdr + regen; }
        else { if (hdr + 1 > bs) return; used = hdr + 1; }
    } else {
        if (bs < 5) return;
        const uint32_t v = zh_ld32(b);
        uint32_t hdr, four, csize;
        if (fmt < 2) { hdr = 3; four = fmt; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
        else if (fmt == 2) { hdr = 4; four = 1; regen = (v >> 4) & 0x3FFF; csize = v >> 18; }
        else { hdr = 5; four = 1; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) + ((uint32_t)b[4] << 10); }
        if (regen > h.blockMax || regen > ZF_BLOCK_MAX) return;
        if (four ? regen < 6 : regen == 0) return;
        if (hdr + csize > bs) return;
        used = hdr + csize;
        if (lt == 2 && csize >= 1) {                                    // a Huffman table of its own; FSE-compressed weights are the serial kind (4-bit ones K1 unpacks by the wave)
            const uint32_t hb = b[hdr];
            if (hb >= 2 && hb < 128 && 1 + hb <= csize) zp_pre_weights(R, S, b + hdr + 1, hb, hdr);
        }
    }
    uint32_t sp = used; const uint32_t send = bs;                        // (relative to the block's first byte)
    if (sp >= send) return;
    uint32_t nbSeq = b[sp++];
    if (nbSeq > 127) {
        if (nbSeq == 255) { if (sp + 2 > send) return; nbSeq = zh_ld16(b + sp) + 0x7F00; sp += 2; }
        else { if (sp >= send) return; nbSeq = ((nbSeq - 128) << 8) + b[sp++]; }
    }
    if (nbSeq == 0 || nbSeq > ZP_SEQ_CAP - 16 || sp >= send) return;
    const uint32_t modes = b[sp++];
    if (modes & 3) return;
    // (K1 parses the three descriptions from an LDS copy of at most 256 bytes: the same bound here, so both read the same bytes)
    const uint32_t hn = send - sp < 256u ? send - sp : 256u;
    const uint8_t* lp = b + sp; const uint8_t* const lend = lp + hn;
    R.seqAt = sp;
#pragma unroll
    for (int kind = 0; kind < 3; kind++) {
        const uint32_t mode = kind == ZD_KIND_LL ? modes >> 6 : kind == ZD_KIND_OF ? (modes >> 4) & 3 : (modes >> 2) & 3;
        if (mode == 3) return;                                          // "repeat" without a dictionary: K1's to refuse
        if (mode == 1) { if (lp >= lend) return; lp += 1; }
        if (mode == 2) {
            uint32_t ms = kind == ZD_KIND_LL ? ZF_MAXLL : kind == ZD_KIND_ML ? ZF_MAXML : ZF_MAXOFF, tl = 0;
            const int r = zd_read_ncount_to(R.norm[kind], lp, lend, &ms, &tl);
            if (r < 0) return;
            R.t[kind].maxSym = ms; R.t[kind].log = tl; R.t[kind].used = (uint32_t)r; R.t[kind].valid = 1;
            lp += r;
        }
    }
}

ZH_DEVFN void zp_pre_body(const ZhipPipeArgs& a, ZpPreLDS& L)
{
    const uint32_t lane = zh_lane();
    for (;;) {
        const uint32_t base = zh_first(zh_atomic_add(a.counters + 11, lane == 0 ? 64u : 0u));
        if (base >= a.count) break;
        const uint32_t i = base + lane;
        if (i < a.count) zp_pre_one(a, a.first + i, a.pre[i], L.lane[lane]);
    }
}

ZH_DEVFN void zp_lit_body(const ZhipPipeArgs& a, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    // zd_build_fse folds the extra-bit counts into its LDS cells: the tables it reads must be initialised (K2 ignores that field)
    if (lane < 36) { L.llBase[lane] = zc_llBase[lane]; L.llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { L.mlBase[lane] = zc_mlBase[lane]; L.mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    bool anyCk = false;                                  // (as zp_lit_lanes_body: KX's word is written once, when the wave leaves)
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counters + 0, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        const uint32_t got1 = zh_first(L.misc[7]);
        zh_sync();
        // dictionary batches (k1Lanes): the frames the lane-per-frame kernel could not finish, listed in `order` (free until KB fills it)
        if (got1 >= (a.k1Lanes ? a.counters[10] : a.count)) { if (anyCk && zh_opaque(lane) == 0) *(volatile uint32_t*)(a.counters + 12) = 1u; break; }
        const uint32_t i = a.k1Lanes ? a.order[got1] : got1;
        const uint32_t f = a.first + i;
        ZdMeta m;
        zp_meta_clear(m);
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
        uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
        const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
        int err = 0; bool fallback = false;
        ZdProf P; P.on = a.prof != nullptr;
        if (P.on) { for (int q = 0; q < ZP_N; q++) P.acc[q] = 0; P.t0 = zd_clock(); }
        do {
            if (srcSize64 > 0x7FFFFFFFull) { fallback = true; break; }
            const uint32_t srcSize = (uint32_t)srcSize64;
            const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
            ZpHdr h;
            err = zp_frame_header(a, src, srcSize, h);
            if (err || h.skippable) break;                                  // (a skippable frame: done, nothing produced)
            const uint32_t blockMax = h.blockMax, hasChecksum = h.hasChecksum;
            const uint64_t fcs = h.fcs;
            uint32_t pos = h.pos;
            m.blockMax = blockMax; m.fcsLo = (uint32_t)fcs; m.fcsHi = (uint32_t)(fcs >> 32);
            if (pos + 3 > srcSize) { err = ZE_SRC_SIZE_WRONG; break; }
            const uint32_t bh = zh_ld24(src + pos); pos += 3;
            const uint32_t lastBlock = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
            if (!lastBlock) { fallback = true; break; }                      // a frame of several blocks (the several-block mode's: zp_lit_mb_body)
            if (type == 3) { err = ZE_CORRUPTION; break; }
            // (block sizes against the frame's maximum: libzstd's one-pass / streaming split, see zd_frame in zhip_decode_kernel.hpp)
            const bool onePass = fcs != ~0ull && cap64 >= fcs;
            if (type < 2) {                                                 // one raw / RLE block: finish right here
                if (type == 0 ? pos + bs > srcSize : pos + 1 > srcSize) { err = ZE_SRC_SIZE_WRONG; break; }
                if (bs > blockMax && !onePass) { err = ZE_CORRUPTION; break; }
                if (bs > cap) { err = ZE_DST_TOO_SMALL; break; }
                if (type == 0) zd_copy_wave(dst, src + pos, bs); else zd_fill_wave(dst, src[pos], bs);
                if (fcs != ~0ull && fcs != bs) { err = ZE_CORRUPTION; break; }
                m.produced = bs;
                if (hasChecksum && a.ckLater) {                             // KX verifies it (a lane per frame; here one lane hashed 128 KiB while 63 waited)
                    const uint32_t cpos = pos + (type == 0 ? bs : 1);
                    if (cpos + 4 > srcSize) { err = ZE_CHECKSUM_WRONG; break; }
                    m.hasChecksum = 1; m.checksum = zh_ld32(src + cpos);
                    anyCk = true;
                } else if (hasChecksum) {
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
            if (bs > blockMax) { err = onePass ? ZE_SRC_SIZE_WRONG : ZE_CORRUPTION; break; }
            if (bs < 2) { err = ZE_CORRUPTION; break; }
            ZdState st;
            st.rep0 = 1; st.rep1 = 4; st.rep2 = 8; st.hufCount = 0; st.llLog = st.ofLog = st.mlLog = 0xFF;
            ZD_T(P, ZP_HEADER);
            ZdLitDefer df; df.table = a.hufTables + (size_t)i * ZP_HUF_CELLS; df.maxLog = ZP_HUF_LOGMAX; df.taken = 0;
            df.log = 0; df.four = 0; df.streamBytes = 0; df.streams = src; df.prevTable = nullptr; df.prevLog = 0; df.shared = 0; df.shareOK = 1;
            df.pre = a.pre ? a.pre + i : nullptr;
            const ZhipDictEntropy* const de = a.dictEntropy;
            if (de) {                                                       // a dictionary with entropy tables: they are the block's "previous" tables
                st.hufCount = de->hufCount;
                if (a.dictTables->hufLog <= ZP_HUF_LOGMAX) { df.prevTable = a.dictTables->huf; df.prevLog = a.dictTables->hufLog; }
                else { zh_sync(); for (uint32_t k = lane; k < 256; k += 64) L.weights[k] = de->hufWeights[k]; zh_sync(); }
            }
            err = zp_block_tables(a, L, st, df, src, pos, bs, blockMax, i, m, P, true, false);
            if (err == (int)ZP_RC_FALLBACK) { err = 0; fallback = true; break; }
            if (err) break;
            if (hasChecksum) {
                if (pos + bs + 4 > srcSize) { err = ZE_CHECKSUM_WRONG; break; }
                m.hasChecksum = 1; m.checksum = zh_ld32(src + pos + bs);
                anyCk = anyCk || a.ckLater != 0;
            }
            m.path = 1;
        } while (false);
        if (fallback) { m.path = 2; }
        if (err) { m.status = err; m.path = 0; }
        zh_sync();
        if (zh_opaque(lane) == 0) {
            if (m.path == 1) zp_enter_bins(a, m);
            a.meta[i] = m;
            if (m.path == 2) { const uint32_t k = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[k] = f; }
            if (m.path == 0) { a.status[f] = m.status; a.outSizes[f] = m.status ? 0 : m.produced; }
        }
#ifdef ZHIP_EMU
        zd_fence();              // (orders the emulator's free-running lanes; on the device the wave is in order and what K1 writes is read by LATER kernels --
#endif                           //  a fence here only made every frame wait out its own stores' round trip: ~2 us of the ~40 a 4 KiB dictionary frame takes, r03l)
        if (P.on) { ZD_T(P, ZP_RAW); if (lane == 0) for (int q = 0; q < ZP_N; q++) if (P.acc[q]) zh_atomic_add64(a.prof + q, P.acc[q]); }
    }
}

// K1 of the several-block mode (ZpFrameRec in zhip_format.hpp): a wave per frame walks the block headers, claims one item per block and
// prepares them in order -- what a block inherits from the one before (Huffman table of a "treeless" literals section, FSE tables in
// "repeat" mode: RFC 8878 3.1.1.3.1.1 / 3.1.1.3.2.1) is this wave's own earlier work: the FSE tables stay in LDS from block to block, the
// Huffman table is copied from the slot of the last block that carried one. Anything unusual about the block LAYOUT (truncation, a
// reserved type, more blocks than item slots) sends the frame to the generic kernel, which answers in stream order like libzstd.
ZH_DEVFN void zp_lit_mb_body(const ZhipPipeArgs& a, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    if (lane < 36) { L.llBase[lane] = zc_llBase[lane]; L.llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { L.mlBase[lane] = zc_mlBase[lane]; L.mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    bool anyCk = false;
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counters + 0, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        const uint32_t i = zh_first(L.misc[7]);
        zh_sync();
        if (i >= a.count) { if (anyCk && zh_opaque(lane) == 0) *(volatile uint32_t*)(a.counters + 12) = 1u; break; }
        const uint32_t f = a.first + i;
        const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
        const uint64_t srcSize64 = a.srcSegs[2 * (size_t)f + 1];
        const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
        ZpFrameRec rec; rec.firstItem = 0; rec.nItems = 0; rec.path = 0; rec.blockMax = 0; rec.fcsLo = rec.fcsHi = 0xFFFFFFFFu; rec.hasChecksum = 0; rec.checksum = 0;
        int err = 0; bool fallback = false;
        ZdProf P; P.on = false;
        do {
            if (srcSize64 > 0x7FFFFFFFull) { fallback = true; break; }
            const uint32_t srcSize = (uint32_t)srcSize64;
            ZpHdr h;
            err = zp_frame_header(a, src, srcSize, h);
            if (err || h.skippable) break;                                  // (a skippable frame: done, nothing produced -- path 0, status 0)
            const uint32_t blockMax = h.blockMax;
            const uint64_t fcs = h.fcs;
            rec.blockMax = blockMax; rec.fcsLo = (uint32_t)fcs; rec.fcsHi = (uint32_t)(fcs >> 32);
            const bool onePass = fcs != ~0ull && cap64 >= fcs;
            // ---- the block layout
            uint32_t nb = 0, end = h.pos;
            for (;;) {
                if (end + 3 > srcSize) { fallback = true; break; }
                const uint32_t bh = zh_ld24(src + end);
                const uint32_t type = (bh >> 1) & 3, bs = bh >> 3, body = type == 1 ? 1u : bs;
                if (type == 3 || end + 3 + body > srcSize || ++nb > ZP_MB_MAXBLOCKS) { fallback = true; break; }
                end += 3 + body;
                if (bh & 1) break;
            }
            if (fallback) break;
            if (h.hasChecksum) {
                if (end + 4 > srcSize) { fallback = true; break; }
                rec.hasChecksum = 1; rec.checksum = zh_ld32(src + end);
                anyCk = anyCk || a.ckLater != 0;
            }
            // ---- nb consecutive items (a frame that does not get them is the generic kernel's; a partly granted range is marked unused)
            zh_sync();
            if (zh_opaque(lane) == 0) L.misc[6] = zh_atomic_add(a.counters + 6, nb);
            zh_sync();
            const uint32_t base = zh_first(L.misc[6]);
            zh_sync();
            if (base + nb > a.itemCap || base + nb < base) {
                ZdMeta z; zp_meta_clear(z);
                for (uint32_t t = base + lane; t < a.itemCap && t - base < nb; t += 64) { a.meta[t] = z; a.itemFrame[t] = i; }
                fallback = true; break;
            }
            rec.firstItem = base; rec.nItems = nb;
            // ---- the blocks
            ZdState st;
            st.rep0 = 1; st.rep1 = 4; st.rep2 = 8; st.hufCount = 0; st.llLog = st.ofLog = st.mlLog = 0xFF;
            const uint16_t* prevTable = nullptr; uint32_t prevLog = 0;
            const ZhipDictEntropy* const de = a.dictEntropy;
            if (de) {                                                       // the dictionary's tables are the first block's "previous" ones
                st.hufCount = de->hufCount;
                if (a.dictTables->hufLog <= ZP_HUF_LOGMAX) { prevTable = a.dictTables->huf; prevLog = a.dictTables->hufLog; }
                zh_sync();
                for (uint32_t k = lane; k < 256; k += 64) L.weights[k] = de->hufWeights[k];
                const uint32_t* T = a.dictTables->fse;
                for (uint32_t k = lane; k < 1280; k += 64) L.fse[k] = T[k];
                st.llLog = de->llLog; st.ofLog = de->ofLog; st.mlLog = de->mlLog;
                zh_sync();
            }
            uint32_t pos = h.pos;
            for (uint32_t j = 0; j < nb; j++) {
                const uint32_t t = base + j;
                ZdMeta m; zp_meta_clear(m);
                const uint32_t bh = zh_ld24(src + pos); pos += 3;
                const uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
                if (err || fallback) { /* a block before this one failed, or the frame already goes to the generic kernel: the item stays unused (and claims nothing) */ }
                else if (type < 2) {
                    if (bs > blockMax && !onePass) err = ZE_CORRUPTION;
                    else { m.path = type == 0 ? 3u : 4u; m.litSize = bs; m.litOff = type == 0 ? pos : (uint32_t)src[pos]; }
                } else if (bs > blockMax) err = onePass ? ZE_SRC_SIZE_WRONG : ZE_CORRUPTION;
                else if (bs < 2) err = ZE_CORRUPTION;
                else {
                    ZdLitDefer df; df.table = a.hufTables + (size_t)t * ZP_HUF_CELLS; df.maxLog = ZP_HUF_LOGMAX; df.taken = 0;
                    df.log = 0; df.four = 0; df.streamBytes = 0; df.streams = src; df.prevTable = prevTable; df.prevLog = prevLog; df.shared = 0; df.shareOK = 0; df.pre = nullptr;
                    err = zp_block_tables(a, L, st, df, src, pos, bs, blockMax, t, m, P, false, true);
                    if (err == (int)ZP_RC_FALLBACK) { err = 0; fallback = true; }       // the chunk's literal room is used up: the generic kernel's frame (its items stay unused)
                    else if (!err) {
                        m.path = 1;
                        const uint32_t lt = src[pos] & 3;                      // literals block type: 2 = a Huffman table of its own, 3 = treeless
                        if (lt >= 2) {
                            // the table a later treeless block inherits: in this item's slot when K1b decodes (df.taken); a table too deep for
                            // the slots was used right here and the next treeless block rebuilds it from the weights still in LDS
                            if (df.taken) { prevTable = df.table; prevLog = df.log; } else { prevTable = nullptr; prevLog = 0; }
                        }
                    }
                }
                if (err && m.status == 0) { m.status = err; m.path = 0; }
                pos += type == 1 ? 1u : bs;
                zh_sync();
                if (zh_opaque(lane) == 0) {
                    if (m.path == 1) zp_enter_bins(a, m);
                    a.meta[t] = m; a.itemFrame[t] = i;
                }
                zh_sync();
            }
            // the first failing block decides the frame's answer -- K3 meets it in stream order (blocks before it may still be refused by K1b / K2)
            err = 0;
            if (fallback) break;
            rec.path = 1;
        } while (false);
        if (fallback) rec.path = 2;
        if (err) rec.path = 0;
        zh_sync();
        if (zh_opaque(lane) == 0) {
            a.frameRecs[i] = rec;
            if (rec.path == 2) { const uint32_t k = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[k] = f; }
            if (rec.path == 0) { a.status[f] = err; a.outSizes[f] = 0; }
        }
#ifdef ZHIP_EMU
        zd_fence();
#endif
    }
}

// ------------------------------------------------------------------------------------------ dictionary tables (one wave, once per dictionary)
ZH_DEVFN void zp_dict_tables_body(const ZhipDictEntropy* de, ZhipDictTables* out, ZdLDS& L)
{
    const uint32_t lane = zh_lane();
    if (lane < 36) { L.llBase[lane] = zc_llBase[lane]; L.llBits[lane] = zc_llBits[lane]; }
    if (lane < 53) { L.mlBase[lane] = zc_mlBase[lane]; L.mlBits[lane] = zc_mlBits[lane]; }
    zh_sync();
    int st = 0;
    for (uint32_t k = lane; k < 256; k += 64) L.weights[k] = de->hufWeights[k];
    zh_sync();
    const int lg = zd_build_huf(L, de->hufCount);
    if (lg < 0) st = ZE_DICT_CORRUPTED;
    else {
        for (uint32_t k = lane; k < (1u << lg); k += 64) out->huf[k] = L.u.huf[k];
        if (lane == 0) out->hufLog = (uint32_t)lg;
    }
    zh_sync();
    L.norm[lane] = lane < 36 ? de->llNorm[lane] : (int16_t)0;
    if (zd_build_fse(L, L.fse + ZD_FSE_LL, de->llMax, de->llLog, ZD_KIND_LL) < 0) st = ZE_DICT_CORRUPTED;
    zh_sync();
    L.norm[lane] = lane < 32 ? de->ofNorm[lane] : (int16_t)0;
    if (zd_build_fse(L, L.fse + ZD_FSE_OF, de->ofMax, de->ofLog, ZD_KIND_OF) < 0) st = ZE_DICT_CORRUPTED;
    zh_sync();
    L.norm[lane] = lane < 53 ? de->mlNorm[lane] : (int16_t)0;
    if (zd_build_fse(L, L.fse + ZD_FSE_ML, de->mlMax, de->mlLog, ZD_KIND_ML) < 0) st = ZE_DICT_CORRUPTED;
    zh_sync();
    for (uint32_t k = lane; k < 1280; k += 64) out->fse[k] = L.fse[k];
    zp_pack_fse(L, (uint32_t*)out->fseK2, de->llLog <= 9 ? de->llLog : 0, de->ofLog <= 8 ? de->ofLog : 0, de->mlLog <= 9 ? de->mlLog : 0);
    if (lane == 0) out->status = st;
    zd_fence();
}

// ------------------------------------------------------------------------------------------ KB (work orders for K2 and K1b)
// Counting sorts by sequence count (K2's order) and by literal count (K1b's), so that lanes sharing a wave run equally long. K1 has done
// the counting: every fast-path frame added itself to its bin's counter and kept the counter's old value as its rank in the bin. What is
// left is a prefix sum over 256 bins (every wave does its own, it is tiny) and one scattered store per frame -- any number of waves.
// (Rounds 1-2 had two waves walk the whole chunk's meta records twice: 0.36-0.49 ms per 32 768 frames, 18 % of a dictionary batch's decode.)
// Blocks [0, gridDim / 2) write K2's order, the others K1b's.
struct ZpBinLDS { uint32_t base[256]; };
ZH_DEVFN void zp_bin_body(const ZhipPipeArgs& a, ZpBinLDS& L)
{
    const uint32_t lane = zh_lane();
    const uint32_t half = zh_nblocks() / 2;
    const bool lit = zh_block() >= half;
    const uint32_t blk = lit ? zh_block() - half : zh_block();
    uint32_t* const order = lit ? a.orderLit : a.order;
    const uint32_t* hist = a.counters + ZP_CNT_BINS + (lit ? 256u : 0u);
    {   const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const uint32_t incl = zh_scan_add(h0 + h1 + h2 + h3), b0 = incl - (h0 + h1 + h2 + h3);
        L.base[4 * lane] = b0; L.base[4 * lane + 1] = b0 + h0; L.base[4 * lane + 2] = b0 + h0 + h1; L.base[4 * lane + 3] = b0 + h0 + h1 + h2;
        if (blk == 0 && lane == 63) a.counters[lit ? 4 : 1] = incl; }
    zh_sync();
    const uint32_t nItems = !a.itemCap ? a.count : a.counters[6] < a.itemCap ? a.counters[6] : a.itemCap;      // several-block mode: the items K1 claimed
    for (uint32_t i = blk * 64 + lane; i < nItems; i += half * 64) {
        const ZdMeta* m = a.meta + i;
        const uint32_t path = m->path, ns = m->nbSeq, lm = m->litMode, ls = m->litSize, rs = m->pad, rl = m->hasChecksum >> 1;
        const uint32_t k = path != 1 ? 0u : lit ? ((lm & 255u) == 3u ? 1u + (ls >> ZP_LITBIN_SHIFT) : 0u) : (ns ? 1u + (ns >> ZP_BIN_SHIFT) : 0u);
        if (k) order[L.base[256 - (k > 256 ? 256u : k)] + (lit ? rl : rs)] = i;
    }
    zd_fence();
}

// ------------------------------------------------------------------------------------------ K1b (Huffman streams, 4 lanes per frame)
// A Huffman stream is a serial chain of table lookups (~150 cycles per symbol) and a frame has at most four of them, so a wave per
// frame keeps 60 of 64 lanes idle (profiles/r01c: 91 % of K1's cycles). Here one wave decodes 16 frames at once -- lane = 4 * slot
// + stream -- with the 16 tables (4 KiB each, built by K1) in LDS; frames come in KB's order so that a wave's streams have similar
// lengths. Same 32-bit bit window as K2: two symbols per v_alignbit, dword refills.
struct alignas(16) ZpVec16 { uint32_t a, b, c, d; };

// ------------------------------------------------------------------------------------------ backward bit reader over an LDS ring
// Both serial chains of the decoder (Huffman streams in K1b, the tANS sequence stream in K2) read a bitstream from its last byte
// down, every lane its own stream at its own pace. Round 1 kept a five-dword queue in registers and shifted it with selects; the load
// issued by one refill was an input of the very next refill's select chain, so every refill waited for a global load issued a few
// hundred cycles earlier (s_waitcnt vmcnt(1) in the loop, profiles/README.md r02a): ~600 cycles of memory latency per refill on a
// chain whose arithmetic is ~100. Here global memory is out of the chain:
//   * every lane owns a ring of RW dwords in LDS, laid out [word][lane] so that no two lanes ever share a bank;
//   * the stream is fetched in aligned 16-byte blocks, up to M per burst, one burst every T loop trips. A block requested in burst b
//     is written to the ring in burst b + 1, i.e. T trips later -- thousands of cycles -- so the wait is free;
//   * a refill is three selects and one ds_read_b32 (the dword after next, consumed a refill later).
// Sizing (C = most bytes a lane can consume between two bursts): a block is requested as soon as it cannot overwrite a dword the
// consumer may still read (block + ring >= cur), so after every burst the requested data reaches below cur - ring; what is readable
// (committed) lags one burst, and the consumer moves at most C per burst: ring >= 2 C + 32 and 16 M >= C + 16 keep it fed whatever
// the input. K2: T = 4 sequences of <= 89 bits, C = 45, ring 128, M = 4. K1b: T = 8 symbols of <= 11 bits, C = 11, ring 64, M = 2.
// Reads may touch the 16-byte blocks around the stream (never another page: blocks are aligned) and never go below the block that
// holds the first byte of the arena side of the stream (offsets are clamped at 0 relative to the aligned base).
#define ZP_NOBLK 0x7FFFFFF0
template <uint32_t RW, uint32_t M, uint32_t LS>
struct ZpBits {
    uint32_t hi, lo, nx, used;      // `used` bits of hi are consumed (1..32 at every group start); lo, nx = the next two dwords
    int32_t cur;                    // offset of lo's dword from p0
    int32_t pb;                     // offset of the next block to request
    int32_t s0;                     // offset of the stream's first byte
    int32_t bo[M]; ZpVec16 blk[M];  // blocks in flight and where they go (ZP_NOBLK: none)
    const uint8_t* p0;              // 16-byte aligned base
    uint32_t* col;                  // this lane's column of the ring: word w of the lane is col[w << LS]
};
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV uint32_t zb_index(int32_t off) { return ((uint32_t)off << (LS - 2)) & ((RW - 1) << LS); }   // off % 4 == 0, LS >= 2
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV ZpVec16 zb_fetch(const ZpBits<RW, M, LS>& B, int32_t off)
{
    return *(const ZpVec16*)(B.p0 + (uint32_t)(off < 0 ? 0 : off));
}
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV void zb_commit(ZpBits<RW, M, LS>& B, int32_t off, const ZpVec16& v)
{
    uint32_t* q = B.col + zb_index<RW, M, LS>(off);      // off is a multiple of 16: four consecutive words, no wrap inside
    q[0] = v.a; q[1u << LS] = v.b; q[2u << LS] = v.c; q[3u << LS] = v.d;
}
// false: empty stream or missing end mark (the caller reports corruption)
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV bool zb_init(ZpBits<RW, M, LS>& B, const uint8_t* p, uint32_t size, uint32_t* col)
{
    B.col = col;
    const uint32_t mis = (uint32_t)((uintptr_t)p & 15);
    B.p0 = p - mis; B.s0 = (int32_t)mis;
#pragma unroll
    for (uint32_t k = 0; k < M; k++) { B.bo[k] = ZP_NOBLK; B.blk[k].a = B.blk[k].b = B.blk[k].c = B.blk[k].d = 0; }
    B.hi = B.lo = B.nx = 0; B.used = 32; B.cur = 0; B.pb = 0;
    if (size == 0) return false;
    const uint32_t last = p[size - 1];
    if (last == 0) return false;
    const int32_t end = (int32_t)(mis + size);
    const int32_t d0 = (end - 1) & ~3, tb = d0 & ~15;
    // the top RW * 4 bytes of the stream, straight into the ring (all loads in flight before the first write)
    ZpVec16 f[RW / 4];
#pragma unroll
    for (uint32_t k = 0; k < RW / 4; k++) f[k] = zb_fetch(B, tb - 16 * (int32_t)k);
#pragma unroll
    for (uint32_t k = 0; k < RW / 4; k++) zb_commit(B, tb - 16 * (int32_t)k, f[k]);
    B.pb = tb - (int32_t)(RW * 4);
    B.hi = B.col[zb_index<RW, M, LS>(d0)]; B.lo = B.col[zb_index<RW, M, LS>(d0 - 4)]; B.nx = B.col[zb_index<RW, M, LS>(d0 - 8)];
    B.cur = d0 - 4;
    B.used = 8 * (uint32_t)(d0 + 4 - end) + 8 - (uint32_t)zh_highbit32(last);      // bytes above the stream + padding + end mark
    return true;
}
// write what the previous burst requested, request what fits now
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV void zb_burst(ZpBits<RW, M, LS>& B)
{
#pragma unroll
    for (uint32_t k = 0; k < M; k++) if (B.bo[k] != ZP_NOBLK) { zb_commit(B, B.bo[k], B.blk[k]); B.bo[k] = ZP_NOBLK; }
#pragma unroll
    for (uint32_t k = 0; k < M; k++) {
        if (B.pb + (int32_t)(RW * 4) >= B.cur) { B.bo[k] = B.pb; B.blk[k] = zb_fetch(B, B.pb); B.pb -= 16; }
    }
}
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV uint32_t zb_top(const ZpBits<RW, M, LS>& B) { return zh_alignbit(B.hi, B.lo, 32u - B.used); }   // the next 32 stream bits
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV void zb_refill(ZpBits<RW, M, LS>& B)                                                         // after `used += n`, n <= 32
{
    const bool ge = B.used > 32;
    B.hi = ge ? B.lo : B.hi; B.lo = ge ? B.nx : B.lo; B.cur = ge ? B.cur - 4 : B.cur; B.used = ge ? B.used - 32 : B.used;
    B.nx = B.col[zb_index<RW, M, LS>(B.cur - 4)];
}
// every bit consumed, no more: the unconsumed part of hi is exactly what lies below the stream's first byte
template <uint32_t RW, uint32_t M, uint32_t LS> ZH_DEV bool zb_finished(const ZpBits<RW, M, LS>& B) { return (int32_t)B.used == 8 * (B.cur + 8 - B.s0); }

#define ZP_HUF_RING 16          // K1b: dwords of ring per lane (64 bytes), 2 blocks per burst, a burst every 8 symbols
typedef ZpBits<ZP_HUF_RING, 2, ZP_HUF_LS> ZpHufBits;
// A decoding cell is 12 bits of information (symbol, code length <= 11): kept as a byte array of symbols and a nibble array of lengths,
// a frame's table is 3 KiB instead of 4, a wave's 16 tables 48 KiB (+ 4 KiB of rings), and THREE waves fit a CU's LDS instead of two --
// the kernel is a latency-bound lookup chain, so residency is throughput.
struct ZpHufLDS { alignas(16) uint8_t sym[ZP_HUF_FRAMES][ZP_HUF_CELLS]; alignas(16) uint8_t len[ZP_HUF_FRAMES][ZP_HUF_CELLS / 2]; uint32_t ring[((ZP_HUF_RING - 1) << ZP_HUF_LS) + 4 * ZP_HUF_FRAMES]; };      // (ring word w of lane l at [w << LS | l], lanes 0 .. 4 x frames - 1: the last row ends with the last lane in use)

ZH_DEV bool zp_huf_stream(const uint8_t* symTab, const uint8_t* lenTab, uint32_t log, const uint8_t* p, uint32_t size, uint8_t* out, uint32_t count,
                          uint32_t* ringCol)
{
    ZpHufBits B;
    if (!zb_init(B, p, size, ringCol)) return false;
#define ZP_LEN(i) (((uint32_t)lenTab[(i) >> 1] >> (((i) & 1) * 4)) & 15u)
#define ZP_PAIR(sa, sb) do { uint32_t top_ = zb_top(B); const uint32_t ia_ = top_ >> sh; sa = symTab[ia_]; const uint32_t la_ = ZP_LEN(ia_); top_ <<= la_; \
        const uint32_t ib_ = top_ >> sh; sb = symTab[ib_]; B.used += la_ + ZP_LEN(ib_); zb_refill(B); } while (0)
    const uint32_t sh = 32 - log;
    uint32_t i = 0;
    // vmcnt counts loads AND stores on gfx9 and the compiler waits conservatively once both are in flight: a store issued right before
    // a burst made the burst's "blocks have arrived" wait sit out the store's own round trip every trip (r02r: the same effect was 17 %
    // of K2). So a trip's eight symbols are stored at the top of the NEXT trip, behind the burst's commits and requests.
    uint64_t pend = 0;
    while (i + 8 <= count) {                    // two symbols (<= 22 bits) per window, eight per trip and per 8-byte store
        zb_burst(B);
#if defined(ZP_K1B_DIAG_NOSTORE)    // DIAGNOSTIC ONLY (wrong literals): K1b without its literal stores
        if (i && pend == 0x0123456789ABCDEFull) zh_st64(out, pend);
#elif defined(ZP_K1B_NT_STORE)
        if (i) __builtin_nontemporal_store(pend, (zh_u64u*)(out + i - 8));
#else
        if (i) zh_st64(out + i - 8, pend);
#endif
        uint32_t s0, s1, s2, s3, s4, s5, s6, s7;
        ZP_PAIR(s0, s1); ZP_PAIR(s2, s3); ZP_PAIR(s4, s5); ZP_PAIR(s6, s7);
        pend = (uint64_t)(s0 | (s1 << 8) | (s2 << 16) | (s3 << 24)) | ((uint64_t)(s4 | (s5 << 8) | (s6 << 16) | (s7 << 24)) << 32);
        i += 8;
    }
    zb_burst(B);                                // the tail (<= 7 symbols) reads what the last burst requested
    if (i) zh_st64(out + i - 8, pend);
    while (i < count) {
        const uint32_t ix = zb_top(B) >> sh;
        B.used += ZP_LEN(ix);
        zb_refill(B);
        out[i++] = symTab[ix];
    }
#undef ZP_PAIR
#undef ZP_LEN
    return zb_finished(B);
}


typedef ZpHufLDS ZpHufKernelLDS;
ZH_DEVFN void zp_huf_body(const ZhipPipeArgs& a, ZpHufKernelLDS& L)
{
    const uint32_t lane = zh_lane(), slot = lane >> 2, strm = lane & 3;
    const uint32_t total = a.counters[4];
    const uint32_t nGroups = (total + ZP_HUF_FRAMES - 1) / ZP_HUF_FRAMES;
    for (;;) {
        const uint32_t g = zh_first(zh_atomic_add(a.counters + 5, lane == 0 ? 1u : 0u));
        if (g >= nGroups) break;
        const uint32_t k = g * ZP_HUF_FRAMES + slot;
        const bool active = slot < ZP_HUF_FRAMES && k < total;
        const uint32_t i = active ? a.orderLit[k] : 0xFFFFFFFFu;
        uint32_t mode = 0, litSize = 0, streamOff = 0, streamBytes = 0;
        if (active) { const ZdMeta* m = a.meta + i; mode = m->litMode; litSize = m->litSize; streamOff = m->litOff; streamBytes = m->produced; }
        const uint32_t log = (mode >> 8) & 255, four = (mode >> 16) & 1;
        zh_sync();
        // the group's tables, HBM -> LDS: whole 4 KiB slots (cells past 2^log are never indexed), 16 bytes per lane, the four loads of a
        // frame in flight before its first LDS write
#pragma unroll 2
        for (uint32_t j = 0; j < ZP_HUF_FRAMES; j++) {
            const uint32_t fj = zh_shfl(i, 4 * j);
            if (fj == 0xFFFFFFFFu) break;                                  // active slots are a prefix
            const bool sharedT = (zh_shfl(mode, 4 * j) & ZP_LIT_SHARED) != 0;      // a treeless block of a dictionary frame: the dictionary's own table
            const ZpVec16* src = sharedT ? (const ZpVec16*)a.dictTables->huf : (const ZpVec16*)(a.hufTables + (size_t)fj * ZP_HUF_CELLS);
            const ZpVec16 r0 = src[lane], r1 = src[lane + 64], r2 = src[lane + 128], r3 = src[lane + 192];
            // eight 2-byte cells (symbol | length << 8) per 16 bytes -> eight symbol bytes + eight length nibbles
#define ZP_SPLIT(v, q) do { const uint32_t w0_ = (v).a, w1_ = (v).b, w2_ = (v).c, w3_ = (v).d; \
                const uint32_t sy0_ = (w0_ & 255) | ((w0_ >> 8) & 0xFF00) | ((w1_ & 255) << 16) | ((w1_ >> 16 & 255) << 24); \
                const uint32_t sy1_ = (w2_ & 255) | ((w2_ >> 8) & 0xFF00) | ((w3_ & 255) << 16) | ((w3_ >> 16 & 255) << 24); \
                const uint32_t ln_ = ((w0_ >> 8) & 15) | ((w0_ >> 24 & 15) << 4) | (((w1_ >> 8) & 15) << 8) | ((w1_ >> 24 & 15) << 12) | \
                                     (((w2_ >> 8) & 15) << 16) | ((w2_ >> 24 & 15) << 20) | (((w3_ >> 8) & 15) << 24) | ((w3_ >> 24 & 15) << 28); \
                ((uint32_t*)L.sym[j])[2 * (lane + 64 * (q))] = sy0_; ((uint32_t*)L.sym[j])[2 * (lane + 64 * (q)) + 1] = sy1_; \
                ((uint32_t*)L.len[j])[lane + 64 * (q)] = ln_; } while (0)
            ZP_SPLIT(r0, 0); ZP_SPLIT(r1, 1); ZP_SPLIT(r2, 2); ZP_SPLIT(r3, 3);
#undef ZP_SPLIT
        }
        zh_sync();
        bool ok = true;
        if (active) {
            const uint32_t f = a.first + (a.itemCap ? a.itemFrame[i] : i);
            const uint8_t* p = a.src + a.srcSegs[2 * (size_t)f] + streamOff;
            uint8_t* lit = a.litArena + (size_t)a.bases[2 * (size_t)i + 1] * 16;
#define ZP_HUF_STREAM(pp, sz, oo, nn) zp_huf_stream(L.sym[slot], L.len[slot], log, pp, sz, oo, nn, L.ring + lane)
            if (!four) { if (strm == 0) ok = ZP_HUF_STREAM(p, streamBytes, lit, litSize); }
            else {
                const uint32_t s1 = zh_ld16(p), s2 = zh_ld16(p + 2), s3 = zh_ld16(p + 4);
                if (6 + s1 + s2 + s3 > streamBytes) ok = false;
                else {
                    const uint32_t seg = (litSize + 3) / 4;
                    const uint32_t so = strm == 0 ? 0 : strm == 1 ? s1 : strm == 2 ? s1 + s2 : s1 + s2 + s3;
                    const uint32_t sz = strm == 0 ? s1 : strm == 1 ? s2 : strm == 2 ? s3 : streamBytes - 6 - s1 - s2 - s3;
                    const uint32_t n = strm < 3 ? seg : litSize - 3 * seg;
                    ok = ZP_HUF_STREAM(p + 6 + so, sz, lit + strm * seg, n);
                }
            }
        }
        const uint64_t badMask = zh_ballot(!ok);
        if (active && strm == 0 && ((badMask >> (4 * slot)) & 15)) {
            ZdMeta* m = a.meta + i;
            m->status = ZE_CORRUPTION; m->path = 0;
            if (!a.itemCap) { const uint32_t f = a.first + i; a.status[f] = ZE_CORRUPTION; a.outSizes[f] = 0; }      // (several-block mode: K3 answers for the frame when it meets the item)
        }
        zd_fence();
        zh_sync();
    }
}

// ------------------------------------------------------------------------------------------ K2q (four lanes == one frame)
// K2 above is bound by the issue rate of ONE wave per CU (r02 SQ counters: 130 instructions per sequence step at 6.5 cycles each; the
// tables of 60 frames fill the CU's LDS, so more waves only split the same frames and every wave still issues the whole step). Here a
// frame is decoded by a QUAD of lanes -- lane 0 the offset stream, 1 match lengths, 2 literal lengths, 3 spare -- so one instruction
// does the cell decode / bit extraction / table lookup of all three tANS streams at once, a wave holds 15 frames and FOUR waves (one per
// SIMD, 15 x 2.5 KiB of tables each) share the CU. What the lanes of a quad owe each other per step -- the bit counts that place every
// field in the stream -- travels by DPP quad_perm, not through LDS.
//   * bit reader: an absolute bit cursor `pos` (identical in the four lanes) over the frame's LDS ring [row][slot]; a field is the two
//     dwords at rows d, d + 1 (row 32 mirrors row 0), one v_alignbit, one v_bfe. No window registers, no refill selects.
//   * the ring is fed like ZpBits: aligned 16-byte blocks requested a burst (4 steps) ahead, lane r of the quad carries block r.
//   * baseline | extra-bit count of the length codes come from one shared LDS word, as in K2 (a per-lane formula needs 7 more instructions
//     than the lookup, and r02l showed this kernel's time is its instruction count).
#ifndef ZQ_FRAMES
#define ZQ_FRAMES 15            // frames per wave; 4 waves x (15 x 2 564 + ring 2 112 + 360) = 163 728 of the CU's 163 840 bytes of LDS
#endif
#define ZQ_ROWS 32              // ring rows (dwords per frame): 128 bytes, as K2's
// (the ring comes first: ds_read2_b32's two offsets are 8 bits each, so only a base within 1 KiB folds into the instruction)
struct ZpSeqQLDS { uint32_t ring[(ZQ_ROWS + 1) * 16]; uint32_t llInfo[36]; uint32_t mlInfo[53]; uint32_t spare; uint8_t tab[ZQ_FRAMES * ZP_K2_STRIDE]; };

// `width` (< 32) bits at absolute bit index q; colAddr = this frame's ring column
ZH_DEV uint32_t zq_field(const uint32_t* col, int32_t q, uint32_t width)
{
    const uint32_t* w = (const uint32_t*)((const uint8_t*)col + (zh_bfe((uint32_t)q, 5, 5) << 6));
    return zh_bfe(zh_alignbit(w[16], w[0], (uint32_t)q), 0, width);
}
ZH_DEV void zq_commit(uint32_t* col, int32_t off, const ZpVec16& v)                // off % 16 == 0: four consecutive rows, no wrap inside
{
    const uint32_t r0 = ((uint32_t)off >> 2) & (ZQ_ROWS - 1);
    uint32_t* q = col + (r0 << 4);
    q[0] = v.a; q[16] = v.b; q[32] = v.c; q[48] = v.d;
    if (r0 == 0) col[ZQ_ROWS << 4] = v.a;
}
ZH_DEV ZpVec16 zq_fetch(const uint8_t* p0, int32_t off) { return *(const ZpVec16*)(p0 + (uint32_t)(off < 0 ? 0 : off)); }

#ifndef ZQ_FENCES
#define ZQ_FENCES 1             // 2: every hand-placed section stays where it is; 1: only "cell request first" (r02q: 6.64 ms against 6.89 / 6.88); 0: the scheduler's order
#endif
#if ZQ_FENCES >= 1
#define ZQ_F1() ZH_SCHED_FENCE()
#else
#define ZQ_F1() do { } while (0)
#endif
#if ZQ_FENCES >= 2
#define ZQ_F2() ZH_SCHED_FENCE()
#else
#define ZQ_F2() do { } while (0)
#endif
// MB: the several-block mode (ZpFrameRec, zhip_format.hpp) -- the item is a block: its frame comes from itemFrame, a block that is not its
// frame's first starts from the SYMBOLIC history, and the history after the block's last sequence is kept for K3 (four more instructions
// per step: a separate instantiation, the single-block kernel is untouched)
template <bool MB>
ZH_DEVFN void zp_seqq_body(const ZhipPipeArgs& a, ZpSeqQLDS& L)
{
    const uint32_t lane = zh_lane(), role = lane & 3, slot = lane >> 2;
    if (lane < 36) L.llInfo[lane] = zc_llBase[lane] | ((uint32_t)zc_llBits[lane] << 24);
    if (lane < 53) L.mlInfo[lane] = zc_mlBase[lane] | ((uint32_t)zc_mlBits[lane] << 24);
    if (lane == 0) L.spare = 512;                    // the spare lane's one-cell "table": symbol 0, x = 512 -> with a 9-bit log no state bits, no extra bits
    zh_sync();
    const bool isOF = role == 0, isSpare = role == 3;
    const uint32_t* const info = role == 1 ? L.mlInfo : L.llInfo;
    const uint32_t tabOff = role == 0 ? ZP_FSE_OF : role == 1 ? ZP_FSE_ML : ZP_FSE_LL;
    const uint32_t total = a.counters[1];
    const uint32_t nGroups = (total + ZQ_FRAMES - 1) / ZQ_FRAMES;
    for (;;) {
        const uint32_t g = zh_first(zh_atomic_add(a.counters + 3, lane == 0 ? 1u : 0u));
        if (g >= nGroups) break;
        const uint32_t k = g * ZQ_FRAMES + slot;
        const bool active = slot < ZQ_FRAMES && k < total;
        const uint32_t i = active ? a.order[k] : 0xFFFFFFFFu;
        const uint32_t logs0 = active ? a.meta[i].logs : 0u;
        zh_sync();
        for (uint32_t j = 0; j < ZQ_FRAMES; j++) {                                 // the group's tables, HBM -> LDS (K1 built them)
            const uint32_t fj = zh_shfl(i, 4 * j);
            if (fj == 0xFFFFFFFFu) break;                                          // active quads are a prefix
            const bool sharedT = (zh_shfl(logs0, 4 * j) & ZP_LOGS_SHARED) != 0;       // every table "repeat": the dictionary's, one copy for all frames
            const uint32_t* src = sharedT ? (const uint32_t*)a.dictTables->fseK2 : (const uint32_t*)(a.fseTables + (size_t)fj * ZP_FSE_CELLS);
            uint32_t* dstw = (uint32_t*)(L.tab + (size_t)j * ZP_K2_STRIDE);
            uint32_t r[ZP_FSE_CELLS / 128];
#pragma unroll
            for (uint32_t q = 0; q < ZP_FSE_CELLS / 128; q++) r[q] = src[lane + 64 * q];
#pragma unroll
            for (uint32_t q = 0; q < ZP_FSE_CELLS / 128; q++) dstw[lane + 64 * q] = r[q];
        }
        uint32_t* const col = L.ring + (active ? slot : 0u);
        const bool idle = isSpare || !active;                                      // idle lanes decode the one-cell table: no bits, the same state for ever
        const uint16_t* const Tm = idle ? (const uint16_t*)&L.spare : (const uint16_t*)(L.tab + (size_t)slot * ZP_K2_STRIDE) + tabOff;
        ZdMeta* const m = a.meta + (active ? i : 0u);
        const uint32_t fi = active ? (MB ? a.itemFrame[i] : i) : 0u;
        const uint32_t f = a.first + fi;
        uint32_t nbSeq = 0, logs = 0;
        int32_t pos = 0, pb = -(1 << 30), fin = 0;
        const uint8_t* p0 = nullptr;
        bool ok = active;
        ZpVec16 v0, v1; v0.a = v0.b = v0.c = v0.d = 0; v1 = v0;
        int32_t tb = 0;
        if (active) {
            const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
            const uint8_t* p = src + m->seqOff; const uint8_t* end = src + m->seqEnd;
            const uint32_t size = (uint32_t)(end - p);
            logs = m->logs;
            const uint32_t lastB = size ? p[size - 1] : 0u;
            if (lastB == 0) ok = false;                                            // empty stream or missing end mark
            else {
                nbSeq = m->nbSeq;
                const uint32_t mis = (uint32_t)((uintptr_t)p & 15);
                p0 = p - mis; fin = (int32_t)(8 * mis);
                const int32_t topByte = (int32_t)(mis + size) - 1;
                tb = topByte & ~15;
                v0 = zq_fetch(p0, tb - 16 * (int32_t)role); v1 = zq_fetch(p0, tb - 16 * (int32_t)(role + 4));   // the top 128 bytes: two blocks per lane
                pb = tb - 128;
                pos = 8 * topByte + zh_highbit32(lastB);
            }
        }
        if (ok) { zq_commit(col, tb - 16 * (int32_t)role, v0); zq_commit(col, tb - 16 * (int32_t)(role + 4), v1); }
        zh_sync();
        const uint32_t llLog = logs & 255, ofLog = (logs >> 8) & 255, mlLog = (logs >> 16) & 255;
        const uint32_t myLog = idle ? 9u : role == 0 ? ofLog : role == 1 ? mlLog : llLog;
        const uint32_t kk = 31 - myLog, size = 1u << myLog;
        // `state` is kept as table index + table size: the update (x << nbBits) + bits lands in [size, 2 size) by construction, so the cell
        // pointer is biased by -size instead of masking every step (the spare lane's one cell answers x = 512, no bits: always index 512)
        const uint16_t* const Tb = Tm - size;
        uint32_t state;
        {   const uint32_t initOff = role == 0 ? llLog : role == 1 ? llLog + ofLog : 0u;       // initial states: LL, OF, ML
            state = idle ? size : size + zq_field(col, pos - (int32_t)(initOff + myLog), myLog);
            pos -= (int32_t)(llLog + ofLog + mlLog); }
        const uint32_t nTrips = zh_first(zh_wave_max(nbSeq)) + 5;                  // trip n produces sequence n - 1 (software pipeline); a group of four is stored at the next group's first trip
        uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
        if (a.dictEntropy) { rep0 = a.dictEntropy->rep[0]; rep1 = a.dictEntropy->rep[1]; rep2 = a.dictEntropy->rep[2]; }     // ZSTD_loadDEntropy's start history
        if (MB && active && a.frameRecs[fi].firstItem != i) { rep0 = ZP_SYM_REP(0); rep1 = ZP_SYM_REP(1); rep2 = ZP_SYM_REP(2); }
        uint32_t fin0 = rep0, fin1 = rep1, fin2 = rep2;              // (MB) the history after the block's LAST sequence (the trips go on to the wave's longest block)
        // (taken into registers HERE: left pending, the loads made the waitcnt pass put a conservative vmcnt wait at the loop's first use of
        // the history -- behind the ring's block load, whose round trip it then sat out every four steps: 5.6 -> 8.2 ms, r02x)
        rep0 = zh_opaque(rep0); rep1 = zh_opaque(rep1); rep2 = zh_opaque(rep2);
        // the offset lane resolves the repeat offsets and stores; the other lanes' stores are parked in the four slots behind the group's first room,
        // so the loop body has no branch. Lanes past their frame's last sequence run on harmlessly: every LDS access is masked, every fetch
        // clamped, and their stores land in the unused tail of the frame's own room (all rooms of a group are the longest frame's size).
        // Sequences leave in groups of four -- 32 aligned bytes, two 16-byte stores (one 8-byte store per step was a partial-sector write
        // each: r02 WRITE_SIZE 242 KB per frame for ~80 KB of sequences). Sequences 4g .. 4g + 3 come out of trips 4g + 1 .. 4g + 4, so a
        // group is stored at the first trip of the NEXT group; the very first store holds nothing and lands in the four slots in front of the room.
        ZpVec16* outp;
        {
            // compact sequence arena: the group claims room for its frames with ONE atomic add -- every frame the group's longest count, rounded to
            // whole store groups, + 4 slots in front (the first, empty store) and + 4 behind (where the other lanes' stores are parked and the last
            // store of a count that is no multiple of four ends): the trips go on to the group's longest frame, so equal rooms need no bound in
            // the loop, and the work order puts frames of like counts together (bins of 128), so little is wasted
            const uint32_t per = ((nTrips - 5 + 3) & ~3u) + 8;
            const uint32_t nAct = (uint32_t)zh_popc64(zh_ballot(active)) >> 2;
            const uint32_t need16 = per * nAct / 2;                                          // (per is a multiple of four: whole 16-byte units)
            const uint32_t room16 = zh_first(zh_atomic_add(a.counters + 7, lane == 0 ? need16 : 0u));      // (sequence rooms grow up from the arena's start ...
            const uint32_t room = room16 * 2;                                                // in sequences from a.seqArena
            if ((uint64_t)room16 + need16 + a.counters[8] > a.arenaBudget16) {               //  ... K1's literal rooms, all claimed by now, down from its end)                               // the chunk's room is used up: the generic kernel's frames
                // (several-block mode: K3 hands the frame over when it meets the item)
                if (active && isOF) { m->path = 2; if (!MB) { const uint32_t q = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[q] = f; } }
                zh_sync();
                continue;
            }
            const uint32_t start = room + slot * per + 4;
            if (active && isOF) a.bases[2 * (size_t)i] = start;
            outp = (ZpVec16*)(a.seqArena + (isOF && active ? (size_t)start - 4 : (size_t)room + per - 4));
        }
        const uint32_t outStep = isOF && active ? 2u : 0u;
        uint32_t g1lo = 0, g1hi = 0, g2lo = 0, g2hi = 0, g3lo = 0, g3hi = 0;       // the group's first three packed sequences
        ZpVec16 w0, w1; w0.a = w0.b = w0.c = w0.d = 0; w1 = w0;                       // the group as stored
        int32_t boff = ZP_NOBLK, posEnd = pos; ZpVec16 blk = v0;
        // The wave is alone on its SIMD and issues one instruction every ~6.5 cycles whatever it is (r02l: time = instructions x steps), so
        // the body is written for instruction count first -- then software-pipelined by hand (as K2 above) so the LDS round trips of the
        // chain are covered: the value / repeat-offset / pack / store work of sequence n - 1 sits right behind the requests of sequence n.
        uint32_t pe0 = 0, pe1 = 0, pqE = 0, pbits = 0, pbase = 1;               // pending pieces of n - 1 (value 1 everywhere: "repeat rep0", a no-op)
        for (uint32_t n0 = 0; n0 < nTrips; n0 += 4) {
            zh_sync();                                       // (every lane is past the previous step's reads: the emulator's lanes are free-running fibers)
            if (boff != ZP_NOBLK) { zq_commit(col, boff, blk); boff = ZP_NOBLK; }
            {   // a block may replace ring bytes [off + 128, off + 144) once nothing at or above them will be read again
                const int32_t t = pb + 128 - (((pos >> 5) << 2) + 4);
                if ((int32_t)(role * 16) <= t) { boff = pb - 16 * (int32_t)role; blk = zq_fetch(p0, boff); }
                int32_t cnt = (t >> 4) + 1; cnt = cnt < 0 ? 0 : cnt > 4 ? 4 : cnt;
                pb -= 16 * cnt; }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t n = n0 + u;
                const uint32_t cell = Tb[state];
                ZQ_F1();
                // ---- sequence n - 1: value, choice of the offset (RFC 8878 3.1.1.5: idx 0 / 1 / 2 = a repeat offset, 3 = rep0 - 1 or a new one)
                const uint32_t val = pbase + zh_bfe(zh_alignbit(pe1, pe0, pqE), 0, pbits);
                const uint32_t mlv = zh_quad<1>(val), llv = zh_quad<2>(val);
                const uint32_t idx = val - 1 + (llv == 0);
                // rep0 - 1 == 0 is no offset: libzstd 1.5.7 forces -1 (zstd.c:46941). It travels as offset 0, which K3 refuses (its range check is on
                // offset - 1: zero wraps to the largest value there is). A new offset beyond the packed form is saturated; K3 knows what to do.
                // (round 4: straight-line selects -- the nested ternary became an exec-mask if / else of 4 scalar instructions + 2 s_nop per step, in a
                // kernel that pays 5.7 cycles for every instruction of its 84-instruction step)
                const uint32_t vm3 = val - 3;
                const uint32_t newOff = vm3 < ZP_OF_LIMIT ? vm3 : ZP_OF_LIMIT;
                const uint32_t c3 = val > 3 ? newOff : rep0 - 1;
                uint32_t offset = idx == 2 ? rep2 : c3; offset = idx == 1 ? rep1 : offset; offset = idx == 0 ? rep0 : offset;
                ZQ_F2();
                // ---- the chain: cell -> bit counts -> where my state bits are -> next state
                const uint32_t sym = cell >> 10, x = cell & 1023;
                const uint32_t inf = info[sym];                                    // baseline | extra-bit count << 24
                const uint32_t nb = (uint32_t)__builtin_clz(x) - kk;
                ZQ_F2();
                // ---- sequence n - 1: history (idx 0: as it was, 1: swap the first two, else: push), pack, store
                rep2 = idx <= 1 ? rep2 : rep1; rep1 = idx == 0 ? rep1 : rep0; rep0 = offset;
                if (MB) {                                                   // trip n has just applied sequence n - 1
                    const bool wasLast = n == nbSeq;
                    fin0 = wasLast ? rep0 : fin0; fin1 = wasLast ? rep1 : fin1; fin2 = wasLast ? rep2 : fin2;
                }
                {   const uint32_t plo = llv | (mlv << 17), phi = (mlv >> 15) | (offset << 3);
                    if (u == 0) {                                                     // (compile-time: the loop is unrolled)
                        w0.a = g1lo; w0.b = g1hi; w0.c = g2lo; w0.d = g2hi; w1.a = g3lo; w1.b = g3hi; w1.c = plo; w1.d = phi;
#if defined(ZQ_DIAG_NOSTORE)        // DIAGNOSTIC ONLY (K3 then reads garbage): K2 without its sequence stores -- what their latency costs the chain (vmcnt is in order)
                        outp += outStep;
#elif defined(ZQ_NT_STORE)
                        {   typedef uint32_t zq_v4 __attribute__((ext_vector_type(4)));
                            zq_v4 t0 = { w0.a, w0.b, w0.c, w0.d }, t1 = { w1.a, w1.b, w1.c, w1.d };
                            __builtin_nontemporal_store(t0, (zq_v4*)outp); __builtin_nontemporal_store(t1, (zq_v4*)(outp + 1)); outp += outStep; }
#else
                        outp[0] = w0; outp[1] = w1; outp += outStep;
#endif
                    } else if (u == 1) { g1lo = plo; g1hi = phi; } else if (u == 2) { g2lo = plo; g2hi = phi; } else { g3lo = plo; g3hi = phi; } }
                ZQ_F2();
                // ---- chain. Fields lie in the stream in the order OF, ML, LL extra bits, then LL, ML, OF state bits: inclusive prefix sums over
                // the quad in two DPP adds each (the spare lane contributes zeros)
                const uint32_t bits = isOF ? sym : inf >> 24;
                const uint32_t P = zh_quad_add<0xCF>(zh_quad_add<0xD3>(bits, bits), bits);      // OF: e0, ML: e0 + e1, LL: e0 + e1 + e2
                const uint32_t R = zh_quad_add<0xFE>(zh_quad_add<0xF9>(nb, nb), nb);            // LL: n2, ML: n2 + n1, OF: n2 + n1 + n0
                const uint32_t totE = zh_quad<2>(P), totN = zh_quad<0>(R);
                const int32_t pE = pos - (int32_t)totE;
                const int32_t qS = pE - (int32_t)R, qE = pos - (int32_t)P;
                const uint32_t* const wS = (const uint32_t*)((const uint8_t*)col + (zh_bfe((uint32_t)qS, 5, 5) << 6));
                const uint32_t* const wE = (const uint32_t*)((const uint8_t*)col + (zh_bfe((uint32_t)qE, 5, 5) << 6));
                const uint32_t s0 = wS[0], s1 = wS[16];
                pe0 = wE[0]; pe1 = wE[16];
                ZQ_F2();
                pqE = (uint32_t)qE; pbits = bits;
                posEnd = n + 1 == nbSeq ? pE : posEnd;         // no state update after the last sequence: the stream must end here
                pos = pE - (int32_t)totN;
                pbase = isOF ? 1u << (sym & 31) : inf & 0xFFFFFFu;
                ZQ_F2();
                state = (x << nb) + zh_bfe(zh_alignbit(s1, s0, (uint32_t)qS), 0, nb);
            }
            // the stores' source registers stay untouched to here: reused as temporaries right behind the store, the write-after-read
            // wait (vmcnt counts in order) also sat out the ring's block load issued just before it -- a load round trip per group (r02x)
            ZH_KEEP4(w0.a, w0.b, w0.c, w0.d); ZH_KEEP4(w1.a, w1.b, w1.c, w1.d);
        }
        // Offsets the packed form cannot hold never stop K2: a new offset of ZP_OF_LIMIT or more is stored AS ZP_OF_LIMIT, "repeat offset 1
        // minus one" = 0 as the largest value there is (r0m1). In a single-block frame both are beyond anything the frame has produced -- K3
        // answers corruption_detected, libzstd's answer (zstd.c:46941, :46558) --, in the several-block mode K3 hands a frame with a saturated
        // offset to the generic kernel. (Watching the largest offset HERE instead sent frames there for nothing: the trips past a block's
        // last sequence decode garbage, and a garbage offset code of 28+ looked like one, r03s.)
        if (active && isOF) {
            const int err = !ok ? ZE_CORRUPTION : posEnd != fin ? ZE_CORRUPTION : 0;
            if (err) { m->status = err; m->path = 0; if (!MB) { a.status[f] = err; a.outSizes[f] = 0; } }
            if (MB) { uint32_t* r = a.itemReps + 4 * (size_t)i; r[0] = fin0; r[1] = fin1; r[2] = fin2; }
        }
        zh_sync();
    }
}

// ------------------------------------------------------------------------------------------ K3 (one wave per frame)
#ifndef ZP_ASM_BYTES
#define ZP_ASM_BYTES ZD_ASM_BYTES       // K3's batch assembly buffer (LDS per wave = this + 1.6 KiB): smaller buffers leave room for a K2 wave beside sixteen K3 waves
#endif
struct ZpExecLDS {
    uint8_t asmb[ZP_ASM_BYTES + 64]; uint16_t mBeg[64]; uint16_t mEnd[64]; uint32_t misc[8];
#if defined(ZP_K3_DIAG_FLOOR) && defined(ZP_FLOOR_LDSPAD)
    uint8_t floorPad[ZP_FLOOR_LDSPAD];               // (diagnostic builds: the LDS a window over the recent output would take, i.e. its occupancy)
#endif
    // the batch's long items (literal runs / far matches above ZD_COOP_LEN bytes), staged together in 16-byte units
    uint16_t uEnd[64], uLit[64], dstL[64], dstM[64], lenL[64], lenM[64]; uint32_t srcL[64], srcM[64];
};

// K3's time is its instruction count (r02 SQ counters: issue-bound, 37 % scalar -- mostly exec-mask bookkeeping of predicated piece loads
// and stores). A piece that is READ from LDS needs none of it: reading past an item is harmless there (the buffer has 64 bytes of slack),
// so four unconditional reads replace the seven predicated ones (r02m: K3 8.68 -> 7.91 ms). The same trick on the store side -- a
// predicated-off store redirected to a per-lane sink slot instead of being branched around -- LOST (9.08 ms: the branchy form skips
// whole stores when no lane of the wave needs them, the sink form always issues all seven).
ZH_DEV void zp_ld32_lds(const uint8_t* q, uint32_t len, uint64_t r[4])              // same contract as zd_ld32: r[3] = the last 8 bytes, or all of them below 8
{
    r[0] = zh_ld64(q); r[1] = zh_ld64(q + 8); r[2] = zh_ld64(q + 16);
    r[3] = zh_ld64(q + (len >= 8 ? len - 8 : 0u));
}

// DICT = false: no dictionary in the context -- every dictionary term folds away (K3 sits at its 128-register cap: carrying the
// dictionary's pointer and size through the dictionary-less kernel spilled 200 bytes per lane and made it 2.6 x slower, r02x)
#ifndef ZP_LIT_SHORT
// K3's registers decide how many waves a SIMD holds, and K3 is short of waves (3 -> 4 per SIMD was x 1.2 in round 2). Items copied by their own
// lane sit in registers between the batch's loads and its LDS stores: 32-byte items = four 8-byte pieces each for the literal run and the far
// match = 16 VGPRs; at 16 bytes (first + last piece) it is 8, longer items ride the 16-byte units, whose pass costs the same for 3 or 30 units.
// Together with the wave-uniform totals in SGPRs that is 128 (+ spills) -> 96 VGPRs = FIVE waves per SIMD: 12.72 -> 10.76 ms per 65 536 frames
// (r03g; six waves would need 80: 64 bytes of spills in the batch loop, 13.6 ms).
#define ZP_LIT_SHORT 16            // literal runs up to this long are copied by their own lane, longer ones as units
#endif
#ifndef ZP_FAR_SHORT
#define ZP_FAR_SHORT 16            // the same for far matches / pre-batch parts staged from global memory
#endif
#ifndef ZP_K3_NT
#define ZP_K3_NT 0              // bit 0: sequences and decoded literals are read with streaming (nt) loads; bit 1: far-match sources too (A/B, r03b)
#endif
#if ZP_K3_NT & 1
#define ZP_SEQ_LD(p) zh_ldq_nt(p)
#define ZP_LIT_LD64(p) zh_ld64_nt(p)
#else
#define ZP_SEQ_LD(p) (*(p))
#define ZP_LIT_LD64(p) zh_ld64(p)
#endif
#if ZP_K3_NT & 2
#define ZP_FAR_LD64(p) zh_ld64_nt(p)
#else
#define ZP_FAR_LD64(p) zh_ld64(p)
#endif
// PROF: the phase timers (ZHIP_PROF=1) are a SEPARATE instantiation. As a run-time flag their eleven 64-bit accumulators lived in VGPRs of the
// production kernel (its 100 SGPRs are taken): 97 -> 77 VGPRs without them, i.e. a sixth wave per SIMD (r03n)
#define ZD_TP(P, i) do { if (PROF) ZD_T(P, i); } while (0)
// what a packed offset stands for once the frame's history at the block's start (R0, R1, R2) is known: itself, or -- several-block mode,
// ZP_SYM_REP in zhip_format.hpp -- entry k of that history minus d; 0xFFFFFFFF = no such offset (libzstd: corruption, zstd.c:46941)
ZH_DEV uint32_t zp_sym_resolve(uint32_t v, uint32_t R0, uint32_t R1, uint32_t R2)
{
    if (v <= ZP_OF_LIMIT) return v;                              // (ZP_OF_LIMIT itself: K2's "too large for the packed form", the caller looks for it)
    if (v > ZP_SYM_TOP) return 0xFFFFFFFFu;
    const uint32_t k = (v - (ZP_OF_LIMIT + 1)) >> 23;
    const uint32_t d = ZP_SYM_REP(k) - v;
    const uint32_t r = k == 0 ? R0 : k == 1 ? R1 : R2;
    return r > d ? r - d : 0xFFFFFFFFu;
}

// One compressed block: the sequences K2 left in slot `t`, the literals of slot `t` (or in place), executed at output position `opRef` of
// the frame at `dst` (MB = false: the frame's only block, position 0). m = the block's record. 0 or a zstd error code.
// LDS hand-offs inside the batch loop: __syncthreads() also waits for the wave's global stores and loads (s_waitcnt vmcnt(0)); -DZP_K3_LIGHT_SYNC
// makes them wave-level fences (the LDS executes a wave's instructions in order), as the round-4 form has them
#define ZP_BSYNC() zh_sync()
#ifndef ZP_HIST_KEEP
#define ZP_HIST_KEEP ((ZP_ASM_BYTES * 5u / 16u) & ~15u)      // bytes of history the buffer keeps when it slides (1 280 of the 4 096) ...
#define ZP_HIST_SLIDE (2u * ZP_HIST_KEEP + 16u)             // ... once more than this many lie in front of the batch (2 576): a batch always finds ZP_ASM_BYTES - ZP_HIST_SLIDE bytes of room
#endif
static_assert((ZP_HIST_SLIDE >= 2 * ZP_HIST_KEEP + 16 || ZP_HIST_KEEP == 0) && ZP_HIST_KEEP % 16 == 0 && ZP_HIST_SLIDE < ZP_ASM_BYTES, "the slide's source and destination must not overlap");      // (-DZP_HIST_KEEP=0u -DZP_HIST_SLIDE=0u: no history, rounds 1-5's form, for A/B)
template <bool DICT, bool PROF, bool MB>
ZH_DEVFN int zp_exec_block(const ZhipPipeArgs& a, ZpExecLDS& L, const ZdMeta& m, uint32_t t, const uint8_t* src, uint8_t* dst, uint32_t cap, uint64_t cap64,
                           uint32_t blockMax, uint32_t& opRef, uint32_t R0, uint32_t R1, uint32_t R2, ZdProf& P)
{
    const uint32_t lane = zh_lane();
    const uint64_t* seqs = a.seqArena + a.bases[2 * (size_t)t];
    const bool litRLE = m.litMode == 2;
    const uint32_t rleByte = m.litOff;
    const uint8_t* litPtr = m.litMode == 0 ? src + m.litOff : a.litArena + (size_t)a.bases[2 * (size_t)t + 1] * 16;
    const uint8_t* const dictEnd = DICT ? a.dictContent + a.dictContentSize : dst;       // position -k of the frame = dictEnd[-k]
    const uint32_t dictSize = DICT ? a.dictContentSize : 0u;
    // Round 6: the assembly buffer keeps what it flushed. A batch is ~0.9 KiB of the buffer's 4: instead of starting every batch at its front, batches are
    // APPENDED -- the hOff bytes in front of the current batch are the block's last output, still in LDS -- and a far match (or the part of a near one that
    // precedes the batch) whose source starts inside that history is staged LDS -> LDS in the staging phase, like a literal run: no memory request, no
    // dependency round (round 3's window was served by the rounds and lost). K3 sits on the memory system's random-request rate (470 M 64-byte read requests
    // per 65 536 frames in 10.1 ms = 47 G/s, TCC_EA0_RDREQ, profiles/r06d_*), and a third of its match sources lie within 2 KiB behind the batch. When the
    // history passes ZP_HIST_SLIDE the last ZP_HIST_KEEP bytes (and the carried tail) move to the buffer's front: two 16-byte copies per lane every other batch.
    uint32_t hOff = 0;                                               // bytes of history in front of the batch (a multiple of 16; wave-uniform)
    uint8_t* asmb = L.asmb;                                          // = L.asmb + hOff: asmb[0] is the byte at position ob, asmb[-k] the byte at ob - k for k <= hOff
    const uint32_t blockStart = MB ? opRef : 0u;
    uint32_t op = blockStart, lp = 0, done = 0;
    // The flush writes whole 16-byte units only: the last `carry` (< 16) bytes of a batch stay at the front of the assembly buffer and leave
    // with the next batch (r03e: the byte-wise tail of every flush -- one lane, up to 15 trips of an LDS read and a byte store -- was a
    // third of the flush phase, 157 K -> 101 K wave-cycles per frame, and made every later store of the frame unaligned). asmb[0] is the byte
    // at absolute position ob = op - carry; every batch-relative offset below (oRel, mRel, mBeg / mEnd, a0 / b0) counts from ob. Bytes
    // below ob are in global memory, bytes from ob on only in LDS: "far" = the whole source lies below ob.
    uint32_t carry = 0;
    const uint32_t nbSeq = m.nbSeq;
    uint64_t qNext = lane < nbSeq ? ZP_SEQ_LD(seqs + lane) : 0;      // the next batch's sequences are requested a batch ahead
    const uint32_t lane0 = lane;
    while (done < nbSeq) {
        const uint32_t avail = nbSeq - done < 64 ? nbSeq - done : 64;
        uint32_t myLL = 0, myML = 0, myOF = 1;
        if (lane < avail) { const uint64_t q = qNext; myLL = ZP_SEQ_LL(q); myML = ZP_SEQ_ML(q); myOF = ZP_SEQ_OF(q); if (MB) myOF = zp_sym_resolve(myOF, R0, R1, R2); }
        if (MB && zh_ballot(lane < avail && myOF == ZP_OF_LIMIT)) return ZP_RC_FALLBACK;      // an offset K2 could not pack: the generic kernel's frame
        uint32_t incL = zh_scan_add(myLL), incT = zh_scan_add(myLL + myML);
        // how many of these fit the assembly buffer (behind the carried bytes)
        const uint64_t fits = zh_ballot(lane < avail && incT + carry <= ZP_ASM_BYTES - hOff);
        uint32_t cnt = (uint32_t)zh_popc64(fits);          // fits is a prefix mask (incT is monotone)
        const bool big = cnt == 0;
        if (big) cnt = 1;
        qNext = done + cnt + lane < nbSeq ? ZP_SEQ_LD(seqs + done + cnt + lane) : 0;
        const bool act = lane < cnt;
        if (!act) { myLL = 0; myML = 0; myOF = 1; }
        // (wave-uniform lane indices: v_readlane into SGPRs -- written as shuffles these totals, and op / lp / carry computed from them, lived in
        // VGPRs of a kernel that sits at its 128-register cap: with the carry the kernel spilled into its batch loop, r03f)
        const uint32_t totL = zh_bcast(incL, cnt - 1), totT = zh_bcast(incT, cnt - 1);
        if (lp + totL > m.litSize) return ZE_CORRUPTION;
        if ((uint64_t)op + totT > cap) return ZE_DST_TOO_SMALL;
        if (op + totT - blockStart > blockMax) return ZE_CORRUPTION;
        const uint32_t litStart = lp + incL - myLL;
        const uint32_t ob = op - carry;
        const uint32_t oRel = incT - (myLL + myML) + carry, mRel = oRel + myLL;
        if (zh_ballot(act && myOF - 1 >= ob + mRel + dictSize)) return ZE_CORRUPTION;      // (offset 0 = K2's "repeat offset 1 minus one = 0": wraps to the largest value; ob < 2^31, dictionary < 2^28)
        if (big) {
            if (lane < carry) dst[ob + lane] = asmb[lane];                  // what the last flush held back
            carry = 0;
            zd_fence();
            const uint32_t bll = zh_bcast(myLL, 0), bml = zh_bcast(myML, 0), bof = zh_bcast(myOF, 0);
            if (litRLE) zd_fill_wave(dst + op, rleByte, bll); else zd_copy_wave(dst + op, litPtr + lp, bll);
            zd_fence();
            zd_match_wave(dst, dictEnd, op + bll, bof, bml);
            zd_fence();
            op += totT; lp += totL; done += 1;
            hOff = 0; asmb = L.asmb;                                            // (the history ends here: this item went straight to memory)
            continue;
        }
        ZD_TP(P, ZP_STAGE);
        const int32_t sAbs = (int32_t)(ob + mRel) - (int32_t)myOF;
        const bool hasM = act && myML > 0;
        const bool farM = hasM && sAbs + (int32_t)myML <= (int32_t)ob;
        uint32_t nearSkip = 0;                                              // bytes of a near match that precede the batch (staged with the far data)
        {
            // Everything this batch reads from global memory is requested before anything is waited for: the short literal runs and
            // short far matches (up to 32 bytes, by their own lanes), then the long items. One at a time by the whole wave, a long item
            // costs a memory round trip each (several per batch: this was most of the kernel's literal / far-match phase); instead
            // every long item of the batch is cut into 16-byte units (the last one shifted back to end with the item), the units are
            // dealt out to the lanes -- a unit finds its item by a binary search over the unit prefix sums -- and the first 64 units'
            // loads fly together with the short ones.
            uint64_t rl[4], rm[4];
            const bool shortL = act && myLL > 0 && myLL <= ZP_LIT_SHORT;
            // the same economy on the global side where reading past the item cannot leave mapped memory: decoded literals live in our own
            // arena (256 bytes of slack per frame); a match source at least 32 bytes below the end of the frame's output slot stays inside it.
            // Lanes without an item read their frame's first bytes. Raw literals (read from the caller's source) keep the exact form.
            const bool arenaLit = m.litMode != 0;                                   // (frame-uniform)
            if (arenaLit && !litRLE) {
                const uint8_t* q = litPtr + (shortL ? litStart : 0u);
                rl[0] = ZP_LIT_LD64(q); rl[3] = ZP_LIT_LD64(q + (shortL && myLL >= 8 ? myLL - 8 : 0u));
                if (ZP_LIT_SHORT > 16) { rl[1] = ZP_LIT_LD64(q + 8); rl[2] = ZP_LIT_LD64(q + 16); } else { rl[1] = 0; rl[2] = 0; }
            } else
            if (shortL && !litRLE) zd_ld32(litPtr + litStart, myLL, rl);
            // a near match whose source starts before the batch: that part is global memory too and is fetched here like a far
            // match (byte by byte in the dependency rounds it was a memory round trip per byte). A lane has one or the other, so
            // both go through ONE pair of piece loads / stores (r02l: K3's time is its instruction count, a pair is ~85 of ~770 per batch)
            const bool pre = hasM && !farM && sAbs < (int32_t)ob;
            const uint32_t preLen = pre ? (uint32_t)((int32_t)ob - sAbs) : 0u;
            nearSkip = preLen;
            const uint32_t lenMi = farM ? myML : preLen;                    // the match item staged here: the whole far match or the part before the batch
            // with a dictionary a source may start below the frame's first byte: wholly there, it is read from the dictionary's content;
            // the rare item that straddles the boundary is copied byte by byte by the whole wave, after the others
            const bool strad = DICT && (farM || pre) && sAbs < 0 && sAbs + (int32_t)lenMi > 0;
            const uint8_t* const mSrc = !DICT || sAbs >= 0 ? dst + sAbs : dictEnd + sAbs;
            const bool shortM = (farM || pre) && lenMi <= ZP_FAR_SHORT && !strad;
            const bool inH = (farM || pre) && sAbs >= (int32_t)(ob - hOff);          // the source starts inside the history: staged from LDS (sAbs >= 0 then)
            if (!zh_ballot(shortM && !inH && sAbs >= 0 && (uint64_t)sAbs + 32 > cap64) && cap64 >= 32) {     // (the dictionary's buffer has its own slack)
                const uint8_t* q = shortM && !inH ? mSrc : dst;
                rm[0] = ZP_FAR_LD64(q); rm[3] = ZP_FAR_LD64(q + (shortM && lenMi >= 8 ? lenMi - 8 : 0u));
                if (ZP_FAR_SHORT > 16) { rm[1] = ZP_FAR_LD64(q + 8); rm[2] = ZP_FAR_LD64(q + 16); } else { rm[1] = 0; rm[2] = 0; }
            } else
            if (shortM && !inH) zd_ld32(mSrc, lenMi, rm);
            if (shortM && inH) { const uint8_t* hq = asmb + (sAbs - (int32_t)ob); rm[0] = zh_ld64(hq); rm[1] = 0; rm[2] = 0; rm[3] = zh_ld64(hq + (lenMi >= 8 ? lenMi - 8 : 0u)); }
            const bool longL = act && myLL > ZP_LIT_SHORT && !litRLE, longM = (farM || pre) && !shortM && !strad;
            const uint32_t uL = longL ? (myLL + 15) >> 4 : 0u, uM = longM ? (lenMi + 15) >> 4 : 0u;
            const uint32_t ue = zh_scan_add(uL + uM);
            const uint32_t U = zh_bcast(ue, 63);
            zh_v16 uv; uv.lo = 0; uv.hi = 0; uint8_t* udp = asmb;
            if (U) {
                L.uEnd[lane] = (uint16_t)ue; L.uLit[lane] = (uint16_t)uL;
                L.srcL[lane] = litStart; L.dstL[lane] = (uint16_t)oRel; L.lenL[lane] = (uint16_t)myLL;
                L.srcM[lane] = (uint32_t)sAbs; L.dstM[lane] = (uint16_t)mRel; L.lenM[lane] = (uint16_t)lenMi;
                ZP_BSYNC();
#define ZP_UNIT(u) do { uint32_t j_ = 0; for (uint32_t stp_ = 32; stp_; stp_ >>= 1) if (L.uEnd[j_ + stp_ - 1] <= (u)) j_ += stp_; \
                    uint32_t k_ = (u) - (j_ ? (uint32_t)L.uEnd[j_ - 1] : 0u); const uint32_t nl_ = L.uLit[j_]; const bool isL_ = k_ < nl_; if (!isL_) k_ -= nl_; \
                    const uint32_t len_ = isL_ ? L.lenL[j_] : L.lenM[j_]; const uint32_t off_ = 16 * k_ + 16 <= len_ ? 16 * k_ : len_ - 16; \
                    const int32_t sm_ = (int32_t)L.srcM[j_]; \
                    if (!isL_ && sm_ >= (int32_t)(ob - hOff)) { const uint8_t* h_ = asmb + (sm_ - (int32_t)ob) + off_; uv.lo = zh_ld64(h_); uv.hi = zh_ld64(h_ + 8); } \
                    else uv = zh_ld128(isL_ ? litPtr + L.srcL[j_] + off_ : (!DICT || sm_ >= 0 ? dst + sm_ : dictEnd + sm_) + off_); \
                    udp = asmb + (isL_ ? L.dstL[j_] : L.dstM[j_]) + off_; } while (0)
                if (lane < U) ZP_UNIT(lane);
            }
            if (litRLE) { for (int k = 0; k < 4; k++) rl[k] = 0x0101010101010101ull * rleByte; }      // (frame-uniform)
            if (shortL) zd_st32(asmb + oRel, myLL, rl);
            if (shortM) zd_st32(asmb + mRel, lenMi, rm);
            if (DICT) for (uint64_t mk = zh_ballot(strad); mk; mk &= mk - 1) {      // dictionary / frame straddlers (at most a few per frame)
                const uint32_t l = (uint32_t)zh_ctz64(mk);
                const uint32_t d = zh_shfl(mRel, l), nn = zh_shfl(lenMi, l);
                const int32_t s0 = (int32_t)zh_shfl((uint32_t)sAbs, l);
                for (uint32_t j = lane; j < nn; j += 64) asmb[d + j] = (uint8_t)zd_hist_byte(dst, dictEnd, s0 + (int32_t)j);
            }
            if (U) {
                if (lane < U) { zh_st64(udp, uv.lo); zh_st64(udp + 8, uv.hi); }
                for (uint32_t u = lane + 64; u < U; u += 64) { ZP_UNIT(u); zh_st64(udp, uv.lo); zh_st64(udp + 8, uv.hi); }
#undef ZP_UNIT
            }
        }
        if (litRLE) {
            for (uint64_t mk = zh_ballot(act && myLL > ZP_LIT_SHORT); mk; mk &= mk - 1) {
                const uint32_t l = (uint32_t)zh_ctz64(mk);
                const uint32_t d = zh_shfl(oRel, l), n = zh_shfl(myLL, l);
                for (uint32_t j = lane; j < n; j += 64) asmb[d + j] = (uint8_t)rleByte;
            }
        }
        // ---- matches that read this batch's own output. A match may start as soon as every near match whose output
        // it reads is done: `need` = the set of those sequences (contiguous index range found by binary search over the
        // batch-relative match extents), so the number of rounds is the dependency depth, not the batch length.
                             // -DZP_K3_NEED_CELLS: the 16-byte cell map below -- measured r04p / r04q: K3 10.09-10.13 ms against 9.95-10.06 (its supersets cost a round now and then; the scalar count rose by what the vector count fell)
        L.mBeg[lane] = (uint16_t)(act ? mRel : 0xFFFF); L.mEnd[lane] = (uint16_t)(act ? mRel + myML : 0xFFFF);
        ZP_BSYNC();
        ZD_TP(P, ZP_EXEC1);
        bool pending = hasM && !farM;
        uint64_t need = 0;
        if (pending) {
            const uint32_t a0 = sAbs > (int32_t)ob ? (uint32_t)(sAbs - (int32_t)ob) : 0;          // first buffer byte I read
            uint32_t b0 = (uint32_t)(sAbs + (int32_t)myML - (int32_t)ob);                          // one past the last byte I read
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
        // a round serves its ready short matches AND every ready long match (one after the other by the whole wave): what is ready at
        // the start of a round never depends on anything else that is ready in it.
        for (;;) {
            const uint64_t pend = zh_ballot(pending);
            if (!pend) break;
            const uint64_t longReady = zh_ballot(pending && myML > ZD_COOP_LEN && (need & ~doneMask) == 0);
            {
                const bool ready = pending && myML <= ZD_COOP_LEN && (need & ~doneMask) == 0;
                if (ready) {
                    // what is left of the match lies in the assembly buffer: source nSrc, destination nSrc + myOF, nLen bytes
                    const uint32_t nLen = myML - nearSkip, nSrc = (uint32_t)(sAbs + (int32_t)nearSkip - (int32_t)ob), nDst = mRel + nearSkip;
                    if (myOF >= nLen) {
                        uint64_t rr[4];
                        zp_ld32_lds(asmb + nSrc, nLen, rr);
                        zd_st32(asmb + nDst, nLen, rr);
                    } else {
                        // the match overlaps its own output (offset < length): the output is periodic with period myOF, so every
                        // step can copy as much as is already final -- the copied length doubles instead of advancing a byte at a time
                        // (what has been written is final and periodic, so the source is simply the `span` bytes before the write
                        // position, span a multiple of the period that doubles while whole spans are copied -- no division)
                        uint32_t done = 0, span = myOF;
                        while (done < nLen) {
                            uint32_t c = span; if (c > nLen - done) c = nLen - done; if (c > 32) c = 32;
                            uint64_t rr[4];
                            zp_ld32_lds(asmb + nDst + done - span, c, rr);
                            zd_st32(asmb + nDst + done, c, rr);
                            done += c;
                            if (c == span) span += span;
                        }
                    }
                    pending = false;
                }
                uint64_t newDone = zh_ballot(ready);
                for (uint64_t lm = longReady; lm; lm &= lm - 1) {
                    const uint32_t pf = (uint32_t)zh_ctz64(lm);
                // whole wave copies one long ready match
                    const uint32_t Frel = zh_shfl(mRel, pf), fml = zh_shfl(myML, pf), fof = zh_shfl(myOF, pf);
                    const int32_t fs = (int32_t)(ob + Frel) - (int32_t)fof;
                    if (fof >= 64) {
                        for (uint32_t c = 0; c < fml; c += 64) {
                            const uint32_t j = c + lane;
                            if (j < fml) { const int32_t sp = fs + (int32_t)j; asmb[Frel + j] = sp >= (int32_t)(ob - hOff) ? asmb[sp - (int32_t)ob] : (DICT ? (uint8_t)zd_hist_byte(dst, dictEnd, sp) : dst[sp]); }
                            if (fof < fml) ZP_BSYNC();
                        }
                    } else {
                        uint32_t idx = lane % fof; const uint32_t adv = 64 % fof;
                        for (uint32_t j = lane; j < fml; j += 64) {
                            const int32_t sp = fs + (int32_t)idx;
                            asmb[Frel + j] = sp >= (int32_t)(ob - hOff) ? asmb[sp - (int32_t)ob] : (DICT ? (uint8_t)zd_hist_byte(dst, dictEnd, sp) : dst[sp]);
                            idx += adv; if (idx >= fof) idx -= fof;
                        }
                    }
                    if (lane == pf) pending = false;
                    newDone |= 1ull << pf;
                    if (lm & (lm - 1)) ZP_BSYNC();
                }
                doneMask |= newDone;
            }
            ZP_BSYNC();
        }
        ZD_TP(P, ZP_EXEC2);
        // vmcnt counts stores as well: the next batch's sequences (requested long ago) are taken into registers HERE, before the flush's
        // stores are issued -- read at the top of the next batch, the wait for them would also sit out the stores just issued
        qNext = zh_opaque64(qNext);
        const uint32_t totB = totT + carry, whole = totB & ~15u;         // bytes in the buffer; the part that leaves now
        {
            // (round 4: the first 1 KiB -- nearly always all of it -- as straight-line code, the loop only behind a uniform test. As one loop the
            // compiler unrolled it by two with eight 4-byte stores per trip and a remainder loop: 63 instructions of the batch's ~700)
            uint8_t* out = dst + ob;
            const uint32_t j0 = lane * 16;
            if (j0 < whole) { const zh_v16 v = *(const zh_v16*)(asmb + j0); *(zh_v16*)(out + j0) = v; }
            if (whole > 1024) for (uint32_t j = j0 + 1024; j < whole; j += 1024) { const zh_v16 v = *(const zh_v16*)(asmb + zh_opaque(j)); *(zh_v16*)(out + j) = v; }
        }
        carry = totB - whole;
        hOff += whole;                                                     // what left stays as history; the tail is where the next batch starts: nothing moves
        ZP_BSYNC();                                                        // (every flush read is done)
        if (hOff > ZP_HIST_SLIDE) {                                        // the last ZP_HIST_KEEP bytes + the tail to the buffer's front (source and destination do not overlap: ZP_HIST_SLIDE >= 2 * ZP_HIST_KEEP + 16)
            const uint32_t s0 = hOff - ZP_HIST_KEEP, n = ZP_HIST_KEEP + carry;
            for (uint32_t j = lane * 16; j < n; j += 1024) { const zh_v16 v = *(const zh_v16*)(L.asmb + s0 + j); *(zh_v16*)(L.asmb + j) = v; }
            hOff = ZP_HIST_KEEP;
            ZP_BSYNC();
        }
        asmb = L.asmb + hOff;
        ZD_TP(P, ZP_FLUSH);
        op += totT; lp += totL; done += cnt;
    }
    if (lane < carry) dst[op - carry + lane] = asmb[lane];               // what the last flush held back
    zd_fence();
    const uint32_t rest = m.litSize - lp;
    if ((uint64_t)op + rest > cap) return ZE_DST_TOO_SMALL;
    if (op + rest - blockStart > blockMax) return ZE_CORRUPTION;
    if (litRLE) zd_fill_wave(dst + op, rleByte, rest); else zd_copy_wave(dst + op, litPtr + lp, rest);
    op += rest;
    opRef = op;
    return 0;
}


// what ends a frame: the content size it announced, its checksum (zstd.c:44264-44277). All lanes call.
ZH_DEVFN int zp_exec_frame_end(ZpExecLDS& L, const uint8_t* dst, uint32_t op, uint32_t fcsLo, uint32_t fcsHi, uint32_t hasChecksum, uint32_t checksum, uint32_t ckLater)
{
    const uint32_t lane = zh_lane();
    const uint64_t fcs = (uint64_t)fcsLo | ((uint64_t)fcsHi << 32);
    if (fcs != ~0ull && fcs != op) return ZE_CORRUPTION;
    if ((hasChecksum & 1) && !ckLater) {                            // (ckLater: KX hashes the frame, a lane per frame -- here ONE lane walked 128 KiB while the wave's other 63 waited: K3 3.1 -> 9.7 ms per 16 384 frames, r06zr)
        zd_fence();
        zh_sync();
        if (zh_opaque(lane) == 0) L.misc[0] = (uint32_t)ze_xxh64(dst, op);
        zh_sync();
        const uint32_t digest = zh_first(L.misc[0]);
        zh_sync();
        if (digest != checksum) return ZE_CHECKSUM_WRONG;
    }
    return 0;
}

#ifdef ZP_K3_DIAG_FLOOR
#include "../../tests/ubench/k3_floor_diag.hpp"
#endif
template <bool DICT, bool PROF>
ZH_DEVFN int zp_exec_frame(const ZhipPipeArgs& a, ZpExecLDS& L, uint32_t i, uint32_t* pProduced, ZdProf& P)
{
    const ZdMeta m = a.meta[i];
    const uint32_t f = a.first + i;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
    uint32_t op = 0;
#ifdef ZP_K3_DIAG_FLOOR          // DIAGNOSTIC ONLY (wrong bytes): the batch loop reduced to its memory traffic -- tests/ubench/k3_floor_diag.hpp
    int e = zp_exec_block_floor<DICT, PROF, false>(a, L, m, i, src, dst, cap, cap64, m.blockMax, op);
#else
    int e = zp_exec_block<DICT, PROF, false>(a, L, m, i, src, dst, cap, cap64, m.blockMax, op, 1, 4, 8, P);
#endif
    if (e) return e;
    e = zp_exec_frame_end(L, dst, op, m.fcsLo, m.fcsHi, m.hasChecksum, m.checksum, a.ckLater);
    if (e) return e;
    *pProduced = op;
    return 0;
}

// a frame of the several-block mode (ZpFrameRec, zhip_format.hpp): its items in order. Raw and RLE blocks are copied / filled here; a
// compressed block's symbolic offsets get their numbers from the history R the blocks before it left, and leave theirs (K2's itemReps).
// An item K1 / K1b / K2 refused ends the frame with that answer -- the first one in stream order, like libzstd; an item K2 could not
// pack sends the frame to the generic kernel (ZP_RC_FALLBACK).
template <bool DICT>
ZH_DEVFN int zp_exec_frame_mb(const ZhipPipeArgs& a, ZpExecLDS& L, uint32_t i, uint32_t* pProduced, ZdProf& P)
{
    const ZpFrameRec rec = a.frameRecs[i];
    const uint32_t f = a.first + i;
    const uint8_t* src = a.src + a.srcSegs[2 * (size_t)f];
    uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
    const uint64_t cap64 = a.dstSegs[2 * (size_t)f + 1];
    const uint32_t cap = cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap64;
    uint32_t R0 = 1, R1 = 4, R2 = 8;
    if (a.dictEntropy) { R0 = a.dictEntropy->rep[0]; R1 = a.dictEntropy->rep[1]; R2 = a.dictEntropy->rep[2]; }
    uint32_t op = 0;
    for (uint32_t bi = 0; bi < rec.nItems; bi++) {
        const uint32_t t = rec.firstItem + bi;
        const ZdMeta m = a.meta[t];
        if (m.path == 0) return m.status ? m.status : ZE_CORRUPTION;
        if (m.path == 2) return ZP_RC_FALLBACK;
        if (m.path >= 3) {
            const uint32_t size = m.litSize;
            if ((uint64_t)op + size > cap) return ZE_DST_TOO_SMALL;
            if (m.path == 3) zd_copy_wave(dst + op, src + m.litOff, size); else zd_fill_wave(dst + op, m.litOff, size);
            zd_fence();
            op += size;
            continue;
        }
        const int e = zp_exec_block<DICT, false, true>(a, L, m, t, src, dst, cap, cap64, rec.blockMax, op, R0, R1, R2, P);
        if (e) return e;
        zd_fence();
        if (m.nbSeq) {
            const uint32_t* r = a.itemReps + 4 * (size_t)t;
            const uint32_t n0 = zp_sym_resolve(r[0], R0, R1, R2), n1 = zp_sym_resolve(r[1], R0, R1, R2), n2 = zp_sym_resolve(r[2], R0, R1, R2);
            R0 = n0; R1 = n1; R2 = n2;
        }
    }
    const int e = zp_exec_frame_end(L, dst, op, rec.fcsLo, rec.fcsHi, rec.hasChecksum, rec.checksum, a.ckLater);
    if (e) return e;
    *pProduced = op;
    return 0;
}

template <bool DICT, bool PROF, bool MB = false>
ZH_DEVFN void zp_exec_body(const ZhipPipeArgs& a, ZpExecLDS& L)
{
    const uint32_t lane = zh_lane();
    // Frames are taken in index order (K2's work order -- longest first -- measured slower, r03f: every wave on a many-sequence frame at the same time
    // makes the far-match gathers of 4 096 waves peak together; the corpus' own mix of heavy and light frames spreads them).
    for (;;) {
        const uint32_t got = zh_atomic_add(a.counters + 2, lane == 0 ? 1u : 0u);
        if (zh_opaque(lane) == 0) L.misc[7] = got;
        zh_sync();
        const uint32_t k = zh_first(L.misc[7]);
        zh_sync();
        if (k >= a.count) break;
        const uint32_t i = k;
        if (MB) { if (a.frameRecs[i].path != 1) continue; }
        else {
        if (a.meta[i].path != 1) continue;
        }
        uint32_t produced = 0;
        ZdProf P; P.on = PROF && a.prof != nullptr;
        if (PROF && P.on) { for (int q = 0; q < ZP_N; q++) P.acc[q] = 0; P.t0 = zd_clock(); }
        int err;
        if constexpr (MB) err = zp_exec_frame_mb<DICT>(a, L, i, &produced, P);
        else err = zp_exec_frame<DICT, PROF>(a, L, i, &produced, P);
        if (PROF && P.on) { ZD_T(P, ZP_RAW); if (lane == 0) for (int q = 0; q < ZP_N; q++) if (P.acc[q]) zh_atomic_add64(a.prof + 16 + q, P.acc[q]); }
        zh_sync();
        if (zh_opaque(lane) == 0) {
            if (MB && err == ZP_RC_FALLBACK) { const uint32_t q = zh_atomic_add(a.fallbackCount, 1u); a.fallbackList[q] = a.first + i;
                                               a.status[a.first + i] = ZE_CORRUPTION; a.outSizes[a.first + i] = 0; }      // (a placeholder the generic kernel overwrites: KX must not take the frame for a finished one)
            else { a.status[a.first + i] = err; a.outSizes[a.first + i] = err ? 0 : produced; }
        }
    }
}

// ------------------------------------------------------------------------------------------ KX (a LANE per frame: content checksums)
// A frame's content checksum is XXH64 over its output (zstd.c:44270-44277) -- four accumulators over 32-byte stripes, a serial chain of 4 096 steps for 128 KiB. K3 (and K1 for raw / RLE
// frames) ran it on ONE lane at the frame's end while the wave's other 63 lanes waited: a batch of checksummed frames took K3 3.1 -> 9.7 ms and K1 0.33 -> 2.4 ms per 16 384 frames
// (`profiles/r06zr_decode_checksum_cost.txt`). Here every lane hashes a frame of its own, after K3: frames that carry a checksum (K1 noted the trailer in the record), were finished by
// the pipeline (path 2 is the generic kernel's, which checks for itself) and decoded without error. A mismatch is the frame's answer, as libzstd's (checksum_wrong, nothing produced).
ZH_DEVFN void zp_check_body(const ZhipPipeArgs& a)
{
    if (a.counters[12] == 0) return;                                // no frame of the chunk carries a checksum
    const uint32_t lane = zh_lane();
    for (uint32_t i = zh_block() * 64 + lane; i < a.count; i += zh_nblocks() * 64) {
        uint32_t has, want, path;
        if (a.itemCap) { const ZpFrameRec& r = a.frameRecs[i]; has = r.hasChecksum & 1u; want = r.checksum; path = r.path; }
        else { const ZdMeta& m = a.meta[i]; has = m.hasChecksum & 1u; want = m.checksum; path = m.path; }
        if (!has || path == 2) continue;
        const uint32_t f = a.first + i;
        if (a.status[f] != 0) continue;
        const uint8_t* dst = a.dst + a.dstSegs[2 * (size_t)f];
        const uint64_t n = a.outSizes[f];
        if ((uint32_t)ze_xxh64(dst, (uint32_t)n) != want) { a.status[f] = ZE_CHECKSUM_WRONG; a.outSizes[f] = 0; }
    }
}
