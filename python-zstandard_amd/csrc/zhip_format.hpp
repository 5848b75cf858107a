// zhip_format.hpp -- zstd format constants (RFC 8878) and error codes shared by host and device code.
#pragma once
#include <stdint.h>

// numeric values = the reference's public enum (zstd/zstd_errors.h:61-97)
enum {
    ZE_OK = 0, ZE_GENERIC = 1, ZE_PREFIX_UNKNOWN = 10, ZE_FRAMEPARAM_UNSUPPORTED = 14, ZE_WINDOW_TOO_LARGE = 16,
    ZE_CORRUPTION = 20, ZE_CHECKSUM_WRONG = 22, ZE_LITERALS_HEADER_WRONG = 24, ZE_DICT_CORRUPTED = 30,
    ZE_DICT_WRONG = 32, ZE_PARAM_UNSUPPORTED = 40, ZE_PARAM_OUTOFBOUND = 42, ZE_TABLELOG_TOO_LARGE = 44, ZE_MAXSYMBOL_TOO_LARGE = 46,
    ZE_MAXSYMBOL_TOO_SMALL = 48, ZE_MEMORY = 64, ZE_DST_TOO_SMALL = 70, ZE_SRC_SIZE_WRONG = 72
};

#define ZF_MAGIC 0xFD2FB528u
#define ZF_MAGIC_SKIPPABLE 0x184D2A50u                    // .. 0x184D2A5F: skippable frames (magic, 4-byte size, payload)
#define ZF_DICT_MAGIC 0xEC30A437u
#define ZF_BLOCK_MAX (1u << 17)
#define ZF_MAXLL 35
#define ZF_MAXML 52
#define ZF_MAXOFF 31
#define ZF_LL_LOGMAX 9
#define ZF_ML_LOGMAX 9
#define ZF_OF_LOGMAX 8

// literal scratch per resident wave: one block's literals (<=128 KiB) + slack for wide stores
#define ZHIP_LIT_STRIDE (ZF_BLOCK_MAX + 256)

// Parsed dictionary entropy section, uploaded once per context (decode side).
// Mirrors what ZSTD_loadDEntropy (zstd.c:44673) extracts: a Huffman table, three FSE distributions, three repcodes.
struct ZhipDictEntropy {
    uint8_t  hufWeights[256];
    uint32_t hufCount;          // number of weights including the implied last one (0 = no entropy tables: raw dict)
    int16_t  ofNorm[32]; uint32_t ofMax, ofLog;
    int16_t  mlNorm[53]; uint32_t mlMax, mlLog;
    int16_t  llNorm[36]; uint32_t llMax, llLog;
    uint32_t rep[3];
    uint32_t contentOffset;     // byte offset of the raw content inside the dictionary blob
    uint32_t dictID;
    int32_t  status;            // 0 or a zstd error code (ZE_DICT_CORRUPTED)
};

// What the decode pipeline takes from a dictionary's entropy section, built ONCE per dictionary (zhip_dict_tables_kernel): the three
// tANS decoding tables in the builder's LDS cell format (K1 drops them into its LDS when a frame's table mode is "repeat") and the
// Huffman decoding table in K1b's cell format (copied into the frame's table slot when its literals are "treeless").
// fseK2: the three tables in K2's 2-byte cell form (symbol << 10 | x), laid out like a frame's slot of the table arena. A frame whose
// three table modes are all "repeat" (every frame of BASELINE configs[3]: 262 144 documents, ONE set of tables) has no slot written at
// all: K2 copies its LDS tables from here, K1b its Huffman table from `huf` (r03k).
struct ZhipDictTables { uint32_t fse[1280]; uint16_t huf[4096]; uint32_t hufLog; int32_t status; uint16_t fseK2[2 * 1280]; };
#define ZP_LOGS_SHARED (1u << 24)       // ZdMeta.logs: the sequence tables are the dictionary's (ZhipDictTables.fseK2), the frame's arena slot is unused
#define ZP_LIT_SHARED (1u << 24)        // ZdMeta.litMode: the Huffman table is the dictionary's (ZhipDictTables.huf)

struct ZhipDecodeArgs {
    const uint8_t* src;             // all frames
    const uint64_t* srcSegs;        // n x (offset, length)
    uint8_t* dst;
    const uint64_t* dstSegs;        // n x (offset, capacity)
    uint64_t* outSizes;             // n
    int32_t* status;                // n
    uint8_t* scratch;               // gridDim.x * ZHIP_LIT_STRIDE
    uint32_t* counter;              // work-stealing counter, zeroed before launch
    uint32_t n;
    uint32_t dictID;
    const uint8_t* dictContent;     // may be null
    uint32_t dictContentSize;
    const ZhipDictEntropy* dictEntropy; // null when no dictionary or raw-content dictionary
    uint64_t maxWindowSize;
    uint32_t magicless;             // 1: frames carry no magic number (ZSTD_f_zstd1_magicless)
    const uint32_t* frameList;      // optional indirection: process frameList[0 .. *listCount) instead of 0 .. n (pipeline fallback)
    const uint32_t* listCount;
    unsigned long long* prof;       // optional: per-phase cycle totals (ZHIP_PROF bring-up / tuning aid), else null
};

// ------------------------------------------------------------------------------------------------ encode side
// per-resident-wave workspace in HBM (hash tables, sequence store, literal buffer, symbol codes)
#define ZE_MAX_HLOG 17
#define ZE_MAX_SEQ 43704                                 // >= 131072 / 3 sequences per block
#define ZE_SEQ_CAP ZE_MAX_SEQ
// packed sequence: offBase[0:28) litLength[28:46) matchLength[46:64)  (lengths <= 128 KiB = one block, offsets < 256 MiB)
#define ZE_SEQ_PACK(off, ll, ml) ((uint64_t)(off) | ((uint64_t)(ll) << 28) | ((uint64_t)(ml) << 46))
#define ZE_SEQ_OFF(q) ((uint32_t)(q) & 0xFFFFFFFu)
#define ZE_SEQ_LL(q) ((uint32_t)((q) >> 28) & 0x3FFFFu)
#define ZE_SEQ_ML(q) ((uint32_t)((q) >> 46))
#define ZE_WS_HASHL 0
#define ZE_WS_HASHS (ZE_WS_HASHL + (4u << ZE_MAX_HLOG))
#define ZE_WS_SEQ   (ZE_WS_HASHS + (4u << ZE_MAX_HLOG))
#define ZE_WS_LIT   (ZE_WS_SEQ + 8u * (ZE_SEQ_CAP + 8))
#define ZHIP_ENC_STRIDE (ZE_WS_LIT + ZF_BLOCK_MAX + 256)

// compression parameter rows, one per source-size class (> 256 KiB, <= 256 KiB, <= 128 KiB, <= 16 KiB): windowLog chainLog hashLog
// searchLog minMatch targetLength strategy, resolved on the host from the level and the caller's explicit parameters (zhip_cparams.hpp);
// the per-source adjustment (ZSTD_adjustCParams_internal) is done per frame on the device
struct ZeRows { int32_t r[4][7]; };

struct ZhipEncodeArgs {
    const uint8_t* src;             // all inputs
    const uint64_t* srcSegs;        // n x (offset, length)
    uint8_t* dst;
    const uint64_t* dstSegs;        // n x (offset, capacity >= compressBound(length))
    uint64_t* outSizes;             // n: frame sizes
    int32_t* status;                // n: 0 or a zstd error code
    uint8_t* workspace;             // gridDim.x * ZHIP_ENC_STRIDE
    uint32_t* counter;              // work-stealing counter, zeroed before launch
    uint32_t n;
    int32_t level;                  // informational (the rows below are what the kernels read)
    uint32_t contentSizeFlag, checksumFlag, dictIDFlag;
    uint32_t magicless;             // 1: frames carry no magic number (ZSTD_f_zstd1_magicless)
    ZeRows rows;
    unsigned long long* prof;
    // two-kernel form (match finding with one LANE per frame, then entropy coding with one wave per frame)
    struct ZeMeta* meta;            // chunk-local per-frame record
    uint8_t* arena;                 // chunk x arenaStride : packed sequences (at 0) + literals (at arenaLit) of each frame
    uint32_t arenaStride, arenaLit; // ZE_ARENA_STRIDE / ZE_ARENA_LIT for 128 KiB sources; dictionary batches (sources below the attach cutoff) use smaller slots
    uint8_t* laneTables;            // (gridDim.x * e1Lanes) x tableStride : hash tables of the frames being searched
    uint32_t e1Lanes;               // frames per wave of the lane-serial match kernel: ZE_E1_LANES (8), ZE_E1_LANES_DICT (32) for dictionary batches
    uint32_t tableStride;
    uint32_t slotSrcMax;            // != 0: table and arena slots of a dictionary batch are sized for sources up to this (the caller's size hint); larger ones are the generic kernel's
    // flat search with an attached dictionary (round 6): its per-document tables are not zeroed any more -- 48 KiB of whole-line stores per 4 KiB document were a seventh of the
    // kernel's line transfers -- but carry the LAUNCH's number above the index: a cell written by an earlier launch reads as empty. tabEpoch == 0: cells are plain indices and
    // the kernel's waves zero the tables first (the emulator's default; dictionaries whose index space leaves fewer than 6 bits); else cell = index | tabEpoch << tabEpochShift,
    // the host zeroes the allocation once and whenever the numbers run out or the index width changes (zhip_compress_batch_device)
    uint32_t tabEpoch, tabEpochShift;
    unsigned long long* waveClock;   // DIAGNOSTIC (null in the product): the flat match kernel's waves leave their duration here, one word per workgroup (-DZHIP_PICK_STUDY=1 reads them)
    uint32_t probeCap;          // != 0: a PROBE launch of the table placement pick (zhip_compress_batch_device) -- the flat search stops after this many bytes of every source; what it leaves is overwritten by the real pass
    uint32_t xxLater;               // != 0: frames leave the entropy kernel with their 4 checksum bytes reserved, not computed -- EX (ze_trailer_body: a LANE per frame) hashes the sources after it
    uint32_t first, count;          // frames [first, first + count) of the batch form this chunk
    // inputs above one block (multi-block frames) are listed by E1 for the generic kernel, which also honours an explicit list
    uint32_t* bigList; uint32_t* bigCount;
    const uint32_t* frameList; const uint32_t* listCount;
    // flat match kernel (one lane per frame, every frame of the chunk in flight): its tables, and the chunk-local indices it
    // leaves to the lane-serial match kernel (strategies / sizes / dictionaries the flat form does not cover)
    uint8_t* flatTables;            // count x tableStride, zeroed by the host before the launch
    uint32_t* e1List; uint32_t* e1Count;
    uint32_t useE1List;             // lane-serial match kernel: 1 = work is e1List[0 .. *e1Count), 0 = the whole chunk
    // sources of several blocks searched by the flat kernel (all null / 0 otherwise; see ZeMbBlock)
    struct ZeMbBlock* mbBlocks;     // count x mbMaxBlocks
    uint32_t* mbCount;              // count : blocks of the frame as the split kernel laid them out; 0 = the generic kernel searches it itself
    uint64_t* mbSeqs;               // count x mbSeqCap packed sequences, block after block
    uint32_t mbMaxBlocks, mbSeqCap;
    uint32_t mbLanes;               // sources per wave of the several-block flat search (<= 64)
    uint32_t mbProbes;              // probes per trip of that search: 2, or 4 for batches bound by a source's serial chain (ze_dfast_flat_np)
    const uint8_t* idle;            // 16 readable bytes the library owns: where the flat searches' lanes without a plausible candidate point their loads (one address per wave)
    // dictionary compression (null / 0 without a dictionary): digested dictionary + its tagged hash tables, all in HBM
    const struct ZeCDict* cdict;
    const uint8_t* cdictContent;
    const uint32_t* cdictHashLong;
    const uint32_t* cdictHashSmall;
};
struct ZeMeta { uint32_t nbSeq, litSize, mode, pad; };      // mode 0: searched (sequences + literals); 1: store raw (too small); 2: error in status;
                                                            // 3: multi-block, listed for the generic kernel; 4: searched, sequences only (flat kernel);
                                                            // 5: multi-block, listed for the generic kernel WITH the flat kernel's sequences (mbBlocks)
// A source of several blocks in the flat match kernel (one lane walks the whole frame: the hash tables and the repeat offsets carry from block
// to block). libzstd decides where a block ends while it compresses -- ZSTD_optimalBlockSize splits a full block only once the blocks before
// have SAVED three bytes (zstd.c:27573), and a block that ends up raw does not pass its repeat offsets on (zstd.c:27361) -- so the search
// runs on an assumption: every block after the first sees savings >= 3, every block is emitted compressed. The split kernel lays the blocks
// out on that assumption, the flat kernel leaves per block where its sequences start, how many, and the repeat offsets after it; the generic
// kernel, which does the entropy coding block by block, checks the assumption against what really happened and searches the frame itself
// where it fails (incompressible data, mostly).
struct ZeMbBlock { uint32_t end, seqStart, nbSeq, rep0, rep1, pad[3]; };
#define ZE_MB_POS_BITS 22                                   // cells of that search: position + 2 in 22 bits, a 10-bit tag (sources below 4 MiB - 8)
#ifndef ZE_FLAT_LANES
#define ZE_FLAT_LANES 64
#endif
#define ZE_ARENA_SEQ 0
#define ZE_ARENA_LIT (8u * (ZE_SEQ_CAP + 8))
#define ZE_ARENA_STRIDE ((size_t)ZE_ARENA_LIT + ZF_BLOCK_MAX + 256)
#ifndef ZE_E1_LANES
#define ZE_E1_LANES 8
#endif
#ifndef ZE_E1_LANES_DICT
#define ZE_E1_LANES_DICT 32             // small uniform sources, one dependent-probe chain each: frames in flight beat divergence (r02i: 8 -> 14.2, 16 -> 17.8, 32 -> 20.5, 64 -> 20.0 GB/s)
#endif
#define ZE_E2_STRIDE ((size_t)ZF_BLOCK_MAX + 256)              // E2 needs one block's literals per resident wave (gathered from the sequences)

// FSE encoding table: per symbol, its cells in table order
struct ZeCTab {
    int32_t log;
    uint32_t maxSym;
    int16_t norm[64];
    uint16_t cellOf[66];
    uint16_t next[512];
};
// Digested compression dictionary (what ZSTD_createCDict keeps, zstd.c:28490-28614), built on the device by
// zhip_build_cdict_kernel. Sources above ZE_DICT_ATTACH_MAX would use the reference's "copy" mode (tables copied, dictionary
// as an external segment); this backend implements the attached mode only and reports larger sources as unsupported.
#define ZE_DICT_ATTACH_MAX (16u * 1024)
#define ZE_DICT_ATTACH_MAX_FAST (8u * 1024)                  // attachDictSizeCutoffs (zstd.c:25250): 8 KB for ZSTD_fast, 16 KB for ZSTD_dfast
#define ZE_CDICT_MAX_HLOG 18
#define ZE_CDICT_MAX_CONTENT ((1u << 24) - ZE_DICT_ATTACH_MAX - 16)   // tagged cells keep 24 bits of index
struct ZeCDict {
    int32_t  status;            // 0 or a zstd error code
    int32_t  hlog, clog, mml;   // parameters the tagged tables were filled with
    int32_t  strat, tlen;       // the dictionary's own strategy (1 fast: one tagged table; 2 double-fast: two) and target length: frames compressed
                                // with an attached dictionary take ALL their parameters but the window from it (ZSTD_resetCCtx_byAttachingCDict)
    uint32_t contentSize, dictID;
    uint32_t rep[3];
    uint32_t hufRepeat, llRepeat, ofRepeat, mlRepeat;    // 0 none, 1 check, 2 valid (HUF_repeat / FSE_repeat)
    uint32_t hufMaxSym;
    uint8_t  hufBits[256];
    uint16_t hufCode[256];
    ZeCTab   tab[3];            // LL, OF, ML
};

// ------------------------------------------------------------------------------------------------ phase-split decode pipeline
// per-frame record handed from kernel to kernel (HBM)
struct ZdMeta {
    int32_t  status;        // 0 or a zstd error code
    uint32_t path;          // 0: finished in K1 (raw / RLE / empty, or failed), 1: fast path (K2 + K3), 2: generic fused kernel
    uint32_t seqOff;        // the sequences bitstream is [seqOff, seqEnd) inside the frame (headers and table descriptions consumed)
    uint32_t seqEnd;
    uint32_t litSize;
    uint32_t litMode;       // 0: literals sit in the frame at litOff, 1: in this frame's literal slot, 2: RLE (litOff = byte),
                            // 3 | log << 8 | fourStreams << 16: Huffman streams at litOff (produced = their byte count) still to be
                            //    decoded into the literal slot by K1b with the table K1 left in the Huffman-table arena
    uint32_t litOff;
    uint32_t nbSeq;         // sequence count (K1 reads it from the sequences header)
    uint32_t blockMax;
    uint32_t fcsLo, fcsHi;  // frame content size (0xFFFFFFFF/0xFFFFFFFF when absent)
    uint32_t produced;
    uint32_t hasChecksum, checksum;   // bit 0: a content checksum to verify after execution (the value); bits 1..: rank in K1b's work-order bin
    uint32_t logs;          // llLog | ofLog << 8 | mlLog << 16 of the frame's FSE tables (built by K1 into the table arena)
    uint32_t pad;           // rank in K2's work-order bin
};
#define ZP_SEQ_CAP 45056u                              // >= 131072 / 3 sequences per block
#define ZP_SEQ_FRONT 16u                               // slots of padding before the first frame's (K2's pipelined store of "sequence -1" lands there)
#define ZP_SEQ_STRIDE ((size_t)ZP_SEQ_CAP * 8)          // packed sequence = ll[0:17) ml[17:35) offset[35:64)
// literal lengths end at 131 071 (code 35: 65 536 + 16 bits), match lengths at 131 074 (code 52: 65 539 + 16 bits) -- 18 bits: a whole 128 KiB
// block can be ONE match into a dictionary (or, in frames of several blocks, into the block before). Offsets of 2^29 and more (windows above
// 512 MiB) do not fit: those frames are the generic kernel's.
#define ZP_SEQ_LL(q) ((uint32_t)(q) & 0x1FFFFu)
#define ZP_SEQ_ML(q) ((uint32_t)((q) >> 17) & 0x3FFFFu)
#define ZP_SEQ_OF(q) ((uint32_t)((q) >> 35))
#define ZP_SEQ_OFBITS 29
#define ZP_LIT_STRIDE ((size_t)ZF_BLOCK_MAX + 256)
#define ZP_LIT_FRONT 256u                               // bytes of padding before the first literal slot (K3 reads the 16 bytes that END with a literal run)
// per-frame FSE decoding tables in HBM / L2: 2-byte cells (symbol << 10 | x), LL 512 + ML 512 + OF 256 cells
#define ZP_FSE_LL 0
#define ZP_FSE_ML 512
#define ZP_FSE_OF 1024
#define ZP_FSE_CELLS 1280
#define ZP_K2_STRIDE 2564                               // K2: LDS bytes of tANS tables per frame: 1280 2-byte cells + 4 (odd dword stride: equal indices never share a bank)
#define ZP_HUF_LOGMAX 11                                // K1b's table slots hold 2^11 2-byte cells (libzstd never emits more; log 12 decodes inside K1)
#define ZP_HUF_CELLS (1u << ZP_HUF_LOGMAX)
#ifndef ZP_HUF_FRAMES
#define ZP_HUF_FRAMES 12                                // frames per K1b wave: 4 lanes (the 4 streams) each; 3 KiB of tables per frame, so 16 -> 3 waves
#endif                                                  // per CU, 12 -> 4, 8 -> 6, 4 -> 12 (48 frames per CU either way; r02c: 8 is 7 % faster than 16 alone, 4 is slower).
                                                        // Round 6: 12 -- ONE wave per SIMD instead of 8's six waves on four SIMDs, and (with the ring trimmed to the lanes in use)
                                                        // 40 896 bytes of LDS, which the 40 932 ONE leaving K2 wave frees still hold: K1b's part beyond K2's end 2.40 -> 1.85 ms (r06zn)
#define ZP_HUF_LS (ZP_HUF_FRAMES > 8 ? 6 : ZP_HUF_FRAMES > 4 ? 5 : 4)     // log2 of the K1b ring's lane stride (dwords): 4 lanes per frame
#define ZP_LITBIN_SHIFT 9                               // K1b work order: frames binned by litSize >> 9
#define ZP_BIN_SHIFT 7                                  // K2 work order: frames binned by nbSeq >> 7 (256 bins), longest first

#define ZP_CNT_BINS 16u                                 // counters[16 .. 16 + 512): the 256 + 256 bin counters of the two work orders
#define ZP_CNT_WORDS (16u + 512u)

// ---- frames of SEVERAL blocks through the same kernels (the chunk's `itemCap` != 0): the work item of K1b / K2 is a BLOCK. K1 (a wave per
// frame) walks the frame's blocks, claims that many consecutive items and fills one ZdMeta each (offsets still relative to the frame's first
// byte; path 3 / 4 = a raw / RLE block: litOff = source offset / the byte, litSize = its size), K3 (a wave per frame) executes the frame's
// items in order. The repeat-offset history crosses block boundaries, K2's items do not wait for each other: the history a later block
// starts with is SYMBOLIC -- three reserved values in the packed offset field standing for "what the frame's history holds here", minus how
// often RFC 8878's "repeat offset 1 minus one" was applied to it -- and K3, which knows the history at each block's start, puts the
// numbers in. K2 leaves each block's final history (same encoding) in itemReps.
struct ZpFrameRec {
    uint32_t firstItem, nItems;
    uint32_t path;              // 1: K3 executes it; 0: finished (status written by K1); 2: the generic kernel's
    uint32_t blockMax;
    uint32_t fcsLo, fcsHi;
    uint32_t hasChecksum, checksum;
};
#define ZP_OF_LIMIT 0x1E000000u                           // real offsets stay below this (K2 sends frames with larger ones to the generic kernel)
#define ZP_SYM_REP(k) (0x1E800000u + ((uint32_t)(k) << 23))   // history entry k of the block's start; minus d = that entry minus d (d < 2^23)
#define ZP_SYM_TOP ZP_SYM_REP(2)                          // values above it (K2's "no such offset" = 0xFFFFFFFF cut to 29 bits) are refused
#define ZP_MB_MAXBLOCKS 4096u                             // frames of more blocks are the generic kernel's
#define ZP_RC_FALLBACK 0x7FFF0002                         // K3's "hand this frame to the generic kernel"
// K0's record of a frame (round 6; zp_pre_body in zhip_decode_pipeline.hpp): what K1's two SERIAL parsers leave behind -- lane 0's walk over the Huffman weights' description
// (distribution, 64-cell FSE table, the weights' two-state decode) and over the three sequence distributions -- made by a LANE per frame one kernel earlier, 64 frames per
// wave, so that K1's wave only builds tables. A part that K0 did not finish (an error, an unusual layout, a table mode that needs no parse) has its count / valid flag at 0
// and K1 parses it itself, as it always did: every error is still found, and worded, by K1's own code.
struct alignas(16) ZpPre {
    uint8_t weights[256];               // the FSE-decoded Huffman weights (zd_read_huf_weights' L.weights[0 .. wCount), before the implied last weight)
    int16_t norm[3][64];                // [ZD_KIND_LL / ZD_KIND_OF / ZD_KIND_ML]: zd_read_ncount's L.norm[0 .. 64)
    uint32_t wCount;                    // 0: weights not made here; else their count
    uint32_t wAt;                       // where the weights' description lies, relative to the block's first byte (K1 uses the record only for THAT description)
    struct { uint32_t valid, maxSym, log, used; } t[3];     // by kind: the distribution's last symbol, table log, bytes used (zd_read_ncount's results)
    uint32_t seqAt;                     // where the first distribution lies, relative to the block's first byte
    uint32_t pad;
};
struct ZhipPipeArgs {
    const uint8_t* src; const uint64_t* srcSegs;
    uint8_t* dst; const uint64_t* dstSegs;
    uint64_t* outSizes; int32_t* status;
    ZdMeta* meta;               // chunk-local
    uint8_t* litArena;          // ONE compact arena of arenaBudget16 x 16 bytes holds literals and sequences (litArena == (uint8_t*)seqArena):
    uint64_t* seqArena;         // a frame's (several-block mode: an item's) literals at litArena + bases[2 i + 1] x 16, its sequences at seqArena + bases[2 i]
    // compact arena (round 5): K1 claims a frame's literal room (regenerated size + 256, the section header says it), K2 a group's sequence room
    // (15 x the group's longest count, rounded, + 8) with one atomic add each (counters[8] / [7]), in 16-byte units of ONE arena -- literals from its
    // end downwards, sequences from its start upwards, so that each kind stays packed --: a block's literals and
    // sequences trade off (every sequence costs three bytes of output or more), so a common budget of 160 KiB per frame covers text (45 + 76 KiB),
    // Huffman-only data (128 KiB + nothing) and match-only data alike; what no longer fits the chunk's budget is the generic kernel's.
    // ~12 GiB of scratch per 65 536-frame chunk instead of 31
    uint32_t* bases;            // chunk (several-block mode: itemCap) x 2
    uint32_t arenaBudget16;
    uint16_t* fseTables;        // chunk x ZP_FSE_CELLS
    uint32_t* order;            // chunk : K2's work list (chunk-local frame indices sorted by decreasing sequence count)
    uint16_t* hufTables;        // chunk x ZP_HUF_CELLS : Huffman decoding tables (symbol | nbBits << 8) for K1b
    uint32_t* orderLit;         // chunk : K1b's work list (frames with Huffman literals, by decreasing literal count)
    uint32_t* counters;         // per chunk slot (ZP_CNT_WORDS words): [0] K1 work, [1] length of `order`, [2] K3 work, [3] K2 group counter,
                                //                 [4] length of `orderLit`, [5] K1b group counter, [6] items claimed (several-block mode),
                                //                 [7] / [8] 16-byte units of the compact arena claimed for sequences (from its start) / for literals (from its end)
    uint32_t* fallbackCount;    // length of the fallback list (shared by every chunk of the batch)
    uint32_t* fallbackList;     // frame indices for the generic kernel
    uint32_t first, count;      // frames [first, first + count) of the batch are this chunk
    // several-block mode (see ZpFrameRec): all null / 0 otherwise -- then item == frame and none of these is read
    uint32_t itemCap;           // item slots of this chunk's arenas (meta, literals, sequences, tables, orders are then per ITEM)
    uint32_t* itemFrame;        // itemCap : the chunk-local frame an item belongs to
    uint32_t* itemReps;         // itemCap x 4 : the history after the block's last sequence, as K2 saw it (symbolic entries possible)
    ZpFrameRec* frameRecs;      // count
    uint64_t maxWindowSize;
    uint32_t magicless;         // 1: frames carry no magic number (ZSTD_f_zstd1_magicless)
    uint32_t ckLater;           // != 0: content checksums are verified AFTER K3 by KX (zp_check_body: a lane per frame) -- K1's raw / RLE frames and K3's frame end only note the trailer;
                                // counters[12] != 0 once a frame of the chunk carries one (KX returns at once on 0)
    ZpPre* pre;                 // chunk : K0's records (null: no K0 ran -- the several-block mode, dictionary batches -- and K1 parses everything itself); counters[11] is K0's work counter
    uint32_t k1Lanes;           // != 0 (dictionary batches, round 6): zhip_decode_lit_lanes_kernel ran first -- a LANE walked every frame whose tables are all the
                                // dictionary's ("treeless" literals, every sequence table "repeat": nothing to build) -- and K1 takes only the frames it listed in `order`
                                // (counters[10] of them; counters[9] is that kernel's work counter)
    unsigned long long* prof;   // optional per-phase cycle totals (ZHIP_PROF tuning aid): [0..9] K1 phases, [16..25] K3 phases
    // dictionary (all null / 0 without one): id, raw content (match sources before the frame's first byte), parsed entropy section, its tables
    uint32_t dictID, dictContentSize;
    const uint8_t* dictContent;
    const ZhipDictEntropy* dictEntropy;     // null for a raw-content dictionary
    const ZhipDictTables* dictTables;       // non-null iff dictEntropy is
};
