"""Kernel LOGIC on the host: the product kernel sources compiled for the wave emulator (tests/emu) against the oracle.
Small inputs only -- the emulator runs each of the 64 lanes as a fiber. The real parity tests are the -m gpu ones."""
import os

import pytest


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    emulib.build()
    return emulib.Emu()


def test_emulated_encoder_is_bit_exact(emu, oracle, corpus):
    raws = [b"", b"foo", b"x" * 7, b"a" * 1000, b"hello world, hello there world! " * 40, bytes(range(256)) * 3]
    raws += [corpus.frame_bytes(i)[: 9000 + 1111 * i] for i in range(6)]
    for flags in (5, 7, 4):
        outs, st = emu.compress_batch(raws, level=3, flags=flags, n_blocks=2)
        assert not any(st)
        for r, o in zip(raws, outs):
            assert o == oracle.compress(r, level=3, flags=flags)


def test_emulated_decoder_matches_oracle(emu, oracle, corpus):
    raws = [b"foo", b"a" * 5000, b"abcabcabcabcabcabcabcabcabc" * 30] + [corpus.frame_bytes(i)[: 20000 + 999 * i] for i in range(6)]
    frames = [oracle.compress(r) for r in raws]
    outs, st = emu.decompress_batch(frames, [len(r) for r in raws], n_blocks=2)
    assert not any(st)
    assert outs == raws
    # a truncated and a bit-flipped frame fail cleanly, the good neighbour still decodes
    bad = [frames[4][: len(frames[4]) // 2], frames[5][:30] + bytes([frames[5][30] ^ 0x40]) + frames[5][31:], frames[3]]
    outs, st = emu.decompress_batch(bad, [len(raws[4]), len(raws[5]), len(raws[3])], n_blocks=1)
    assert st[0] != 0 and st[2] == 0 and outs[2] == raws[3]


def test_emulated_decoder_on_golden_libzstd_frames(emu):
    import hashlib
    from tests.test_oracle_vs_golden import GOLD, BLOB
    frames, sizes, shas = [], [], []
    for case in GOLD["cases"]:
        rec = case["frames"]["default"]
        if "blob_offset" in rec and case["size"] <= 20000:
            frames.append(BLOB[rec["blob_offset"]: rec["blob_offset"] + rec["size"]]); sizes.append(case["size"]); shas.append(case["input_sha256"])
    outs, st = emu.decompress_batch(frames, sizes, n_blocks=2)
    assert not any(st) and len(frames) >= 8
    assert [hashlib.sha256(o).hexdigest() for o in outs] == shas


def test_emulated_dictionary_compression_matches_golden(emu):
    """device dictionary digestion (entropy tables, tagged hash tables) + attached-dictionary search, fused and two-kernel forms"""
    import hashlib
    from tests.test_oracle_vs_golden import GOLD, _dict_vectors
    dicts, srcs = _dict_vectors()
    for key, flags in (("trained/default", 5), ("raw/default", 5), ("trained/checksum_nodictid", 3)):
        name = key.split("/")[0]
        for pipeline in (False, True):
            outs, st = emu.compress_batch(srcs, level=3, flags=flags, n_blocks=2, pipeline=pipeline, dict_data=dicts[name])
            assert not any(st)
            for o, rec in zip(outs, GOLD["dictionary_compress"]["frames"][key]):
                assert len(o) == rec["size"] and hashlib.sha256(o).hexdigest() == rec["sha256"], (key, pipeline)
    outs, st = emu.compress_batch([b"a" * ((1 << 19) + 1), b"abc" * 100], level=1, flags=5, n_blocks=1, pipeline=True, dict_data=dicts["raw"])
    assert st[0] == 40 and st[1] == 0       # parameter_unsupported only for the source whose window (level 1: 512 KiB) would drop the dictionary


def test_emulated_decode_pipeline_matches_oracle(emu, oracle, corpus):
    """K1 -> bin -> K2 -> K3 under emulation: mixed frame kinds in one batch, several chunkings, errors isolated per frame"""
    import numpy as np
    from tests.test_oracle_vs_golden import GOLD, BLOB
    rng = np.random.default_rng(9)
    raws = [b"", b"foo", b"a" * 5000, b"abcabcabcabcabcabcabcabcabc" * 300, rng.bytes(3000), bytes(rng.integers(0, 4, 9000, dtype=np.uint8))]
    raws += [corpus.frame_bytes(i)[: 6000 + 2777 * i] for i in range(10)]
    frames = [oracle.compress(r, flags=7 if k % 3 == 0 else 5) for k, r in enumerate(raws)]
    mb = GOLD["multiblock_level19"]                      # multi-block frame: takes the fallback list to the generic kernel
    frames.append(BLOB[mb["blob_offset"]: mb["blob_offset"] + mb["frame_size"]]); raws.append(oracle.decompress(frames[-1], mb["size"]))
    sizes = [len(r) for r in raws]
    for chunk in (0, 5):
        outs, st, nfb = emu.decompress_pipeline(frames, sizes, n_blocks=3, chunk=chunk)
        assert not any(st) and nfb == 1
        assert outs == raws
    # damage: truncated, bit-flipped in the sequences section, wrong checksum -- neighbours unaffected
    bad = list(frames[:-1])
    bad[7] = bad[7][: len(bad[7]) - 9]
    b = bytearray(bad[9]); b[-20] ^= 0x10; bad[9] = bytes(b)
    b = bytearray(bad[6]); b[-1] ^= 0xFF; bad[6] = bytes(b)          # frame 6 carries a checksum (k % 3 == 0)
    outs, st, nfb = emu.decompress_pipeline(bad, sizes[:-1], n_blocks=2, chunk=4)
    assert st[7] != 0 and st[6] == 22
    for k in range(len(bad)):
        if k not in (6, 7, 9):
            assert st[k] == 0 and outs[k] == raws[k]
    if st[9] == 0:
        assert outs[9] == oracle.decompress(bad[9], sizes[9])
    # damage inside the Huffman streams / tree description (K1, K1b): never accept what the oracle rejects
    for pos in (14, 40, 90, 200, 400):
        bad = list(frames[:-1])
        for k in (10, 12, 14):
            b = bytearray(bad[k]); b[pos] ^= 0x04; bad[k] = bytes(b)
        outs, st, nfb = emu.decompress_pipeline(bad, sizes[:-1], n_blocks=2, chunk=0)
        for k in (10, 12, 14):
            try:
                want = oracle.decompress(bad[k], sizes[k])
            except RuntimeError:
                want = None
            if want is None:
                assert st[k] != 0, (pos, k)
            elif st[k] == 0:
                assert outs[k] == want, (pos, k)
        assert st[11] == 0 and outs[11] == raws[11]


def test_emulated_fast_strategy_matches_golden(emu, corpus):
    """device `fast` parser (levels 1, 2, negative) in the fused and the two-kernel form, small inputs of the golden set"""
    import hashlib
    from tests.test_oracle_vs_golden import GOLD, _inputs
    inputs = _inputs()
    names = [n for n in GOLD["levels"]["frames"] if len(inputs[n]) <= 20000]
    assert len(names) >= 20
    raws = [inputs[n] for n in names]
    for lvl in (1, 2, -5):
        for pipeline in (False, True):
            outs, st = emu.compress_batch(raws, level=lvl, flags=5, n_blocks=2, pipeline=pipeline)
            assert not any(st)
            for n, o in zip(names, outs):
                rec = GOLD["levels"]["frames"][n][str(lvl)]
                assert len(o) == rec["size"] and hashlib.sha256(o).hexdigest() == rec["sha256"], (n, lvl, pipeline)


def test_emulated_multiblock_encode_matches_golden(emu):
    """inputs above one block are listed by E1 and encoded by the generic kernel: block split, carried state, raw / RLE blocks"""
    import hashlib
    from tests.test_oracle_vs_golden import GOLD, _multiblock_inputs
    data = _multiblock_inputs()
    names = ["random1M", "corpus128k+1", "corpus300k", "rle_tail"]
    raws = [data[n] for n in names] + [b"tiny input next to the big ones " * 20]
    for lvl, pipeline in ((3, True), (1, False)):
        outs, st = emu.compress_batch(raws, level=lvl, flags=5, n_blocks=2, pipeline=pipeline)
        assert not any(st)
        for n, o in zip(names, outs):
            want = GOLD["multiblock_compress"]["frames"][n]["%d/default" % lvl]
            assert len(o) == want["size"] and hashlib.sha256(o).hexdigest() == want["sha256"], (n, lvl)


def _flat_search_inputs(corpus, seed=77, count=36):
    import numpy as np
    rng = np.random.default_rng(seed)
    raws = []
    for i in range(count):
        kind = i % 9
        n = int(rng.integers(64, 131073)) if i % 4 else int(rng.integers(1, 4000))
        if kind == 0: r = corpus.frame_bytes(int(rng.integers(0, 500)))[:n]
        elif kind == 1: r = rng.bytes(n)
        elif kind == 2: r = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        elif kind == 3: r = (b"abcdefgh" * (n // 8 + 1))[:n]
        elif kind == 4: r = (rng.bytes(700) * (n // 700 + 1))[:n]
        elif kind == 5:
            a = bytearray(corpus.frame_bytes(int(rng.integers(0, 500)))[:n])
            for k in range(0, len(a), 997): a[k] = int(rng.integers(0, 256))
            r = bytes(a)
        elif kind == 6: r = b"\0" * n
        elif kind == 7:
            parts, tot = [], 0
            while tot < n:
                m = int(rng.integers(1, 5000))
                parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0, 256))]) * m); tot += m
            r = b"".join(parts)[:n]
        else:
            blk = rng.bytes(300); r = (blk + rng.bytes(40000) + blk * 3 + rng.bytes(50000) + blk)[:n]   # offsets beyond 64 KiB
        raws.append(r)
    raws += [corpus.frame_bytes(7), b"x" * 63, b"y" * 64, b"hello " * 11]
    return raws


def test_emulated_flat_match_kernel_and_wave_entropy_coder(emu, oracle, corpus):
    """two-kernel form at level 3: the flat double-fast kernel (one lane per frame, tagged cells, two probes per trip, sequences
    only) + the wave-parallel entropy kernel (literals gathered from the sequence list, 16-lane Huffman streams, 3-lane tANS
    chains) against the oracle on inputs that reach every branch: long literal runs (> 63), long matches (> 130), far and
    repeated offsets, matches that run into the end of the input, incompressible and single-byte inputs, tiny inputs that the
    flat kernel hands to the lane-serial kernel (< 64 bytes), full 128 KiB blocks"""
    import ctypes
    import numpy as np
    raws = _flat_search_inputs(corpus)
    emu.lib.emu_stat.restype = ctypes.c_long
    before = emu.lib.emu_stat(15)
    for flags in (5, 7):
        outs, st = emu.compress_batch(raws, level=3, flags=flags, n_blocks=3, pipeline=True, chunk=17)
        assert not any(st)
        for i, (r, o) in enumerate(zip(raws, outs)):
            assert o == oracle.compress(r, level=3, flags=flags), (flags, i, len(r))
    assert emu.lib.emu_stat(15) - before >= 2 * 30, "the flat match kernel did not take these frames"
    # small batches: the same search with the frame's source copied to LDS by its own wave (ze_match_lds_body) -- sizes around the
    # 16-byte copy units, the frames it hands on (tiny, above one block), chunks on both sides of the switch
    raws += [corpus.frame_bytes(9)[:n] for n in (64, 65, 79, 80, 81, 4095, 131071)] + [corpus.frame_bytes(9) + b"tail past one block"]
    emu.lib.emu_set_e1lds_max(24)
    try:
        before = emu.lib.emu_stat(15)
        for chunk in (17, 30):
            outs, st = emu.compress_batch(raws, level=3, flags=5, n_blocks=3, pipeline=True, chunk=chunk)
            assert not any(st)
            for i, (r, o) in enumerate(zip(raws, outs)):
                assert o == oracle.compress(r, level=3, flags=5), (chunk, i, len(r))
        assert emu.lib.emu_stat(15) - before >= 2 * 36
        # a smaller LDS area (the shape of a batch of small sources); a source above it is searched in place, same frame
        emu.lib.emu_set_e1lds_bytes(4096)
        outs, st = emu.compress_batch(raws, level=3, flags=5, n_blocks=3, pipeline=True, chunk=17)
        assert not any(st) and all(o == oracle.compress(r, level=3, flags=5) for r, o in zip(raws, outs))
    finally:
        emu.lib.emu_set_e1lds_max(0); emu.lib.emu_set_e1lds_bytes(131072)


def test_emulated_four_probe_flat_search(emu, oracle, corpus):
    """The flat double-fast search with FOUR probes per trip (ze_dfast_flat_np, round 4: the form chunks of up to 32 768 sources take -- they are bound
    by a source's serial chain, and a trip of four speculative probes consumes 2.95 of them on average instead of 1.8): the same inputs as the
    two-probe kernel's test, through the flat kernel and through the LDS-source kernel, byte for byte the oracle's frames; and fewer trips."""
    import ctypes
    raws = _flat_search_inputs(corpus) + _flat_search_inputs(corpus, seed=11, count=27)
    emu.lib.emu_stat.restype = ctypes.c_long
    trips = {}
    try:
        for probes in (2, 3, 4):
            emu.lib.emu_set_probes(probes)
            for ldsmax in ((0,) if probes == 3 else (0, 64)):        # (three probes: the flat kernel's launches of 32 769 ... 65 536 sources only)
                emu.lib.emu_set_e1lds_max(ldsmax)
                t0 = emu.lib.emu_stat(10)
                outs, st = emu.compress_batch(raws, level=3, flags=5, n_blocks=3, pipeline=True, chunk=40)
                assert not any(st)
                for i, (r, o) in enumerate(zip(raws, outs)):
                    assert o == oracle.compress(r, level=3, flags=5), (probes, ldsmax, i, len(r))
                trips[(probes, ldsmax)] = emu.lib.emu_stat(10) - t0
    finally:
        emu.lib.emu_set_probes(2); emu.lib.emu_set_e1lds_max(0)
    assert trips[(4, 0)] < 0.75 * trips[(2, 0)] and trips[(4, 64)] < 0.75 * trips[(2, 64)] and trips[(4, 0)] < trips[(3, 0)] < trips[(2, 0)], trips


def test_computed_sequence_codes_match_the_format_tables(emu):
    """the entropy kernel computes LL / ML codes and extra-bit counts instead of reading tables: every length up to one block"""
    assert emu.lib.emu_check_code_formulas() == 0


def test_wave_parallel_table_builders_match_the_serial_restatement(emu, oracle, corpus):
    """The entropy kernel builds its FSE tables (normalisation incl. the fallback distribution, the table description, the encoding
    table) and its Huffman code with one lane per symbol / cell; the oracle does the same serially, the way libzstd does
    (zstd.c:16402, :16316, :16170, :16005, :17513). Same histograms in, same tables out -- on far more shapes than whole frames reach."""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(1)
    e, o = emu.lib, oracle.lib
    done = 0
    for it in range(1500):
        max_sym = int(rng.integers(1, 53))
        kind = it % 4
        if kind == 0: cnt = rng.integers(0, 50, max_sym + 1)
        elif kind == 1: cnt = (rng.pareto(1.0, max_sym + 1) * 20).astype(np.int64)
        elif kind == 2: cnt = rng.integers(0, 3, max_sym + 1) * rng.integers(1, 2000, max_sym + 1)
        else: cnt = rng.integers(0, 5000, max_sym + 1)
        cnt = cnt.astype(np.uint32); cnt[max_sym] = max(1, cnt[max_sym])
        if np.count_nonzero(cnt) < 2: cnt[0] = 3
        total = int(cnt.sum())
        nz = int(np.count_nonzero(cnt))
        if cnt.max() == total: continue
        lg = int(rng.integers(max(5, int(np.ceil(np.log2(nz))) + 1), 10))
        if lg > 9 or (1 << lg) < nz: continue
        outs = []
        for lib, fn in ((e, "emu_fse_tables"), (o, "zo_test_fse_tables")):
            norm = np.zeros(64, dtype=np.int16); nc = np.zeros(512, dtype=np.uint8); h = C.c_uint32(0)
            cell_of = np.zeros(66, dtype=np.uint16); nxt = np.zeros(512, dtype=np.uint16)
            rc = getattr(lib, fn)(cnt.ctypes.data_as(C.c_void_p), C.c_uint32(max_sym), C.c_uint32(total), C.c_uint32(lg), C.c_int(int(total >= 2048)),
                                  norm.ctypes.data_as(C.c_void_p), nc.ctypes.data_as(C.c_void_p), C.byref(h), cell_of.ctypes.data_as(C.c_void_p), nxt.ctypes.data_as(C.c_void_p))
            outs.append((rc, norm[: max_sym + 1].tolist(), nc[: h.value].tobytes(), cell_of[: max_sym + 2].tolist(), nxt[: 1 << lg].tolist()) if rc == 0 else (rc,))
        assert outs[0] == outs[1], (it, max_sym, lg, cnt.tolist())
        done += 1
    assert done > 1000
    done = 0
    for it in range(400):
        kind = it % 5
        if kind == 0: data = np.frombuffer(corpus.frame_bytes(it)[: int(rng.integers(300, 131072))], dtype=np.uint8)
        elif kind == 1: data = rng.integers(0, int(rng.integers(2, 256)), int(rng.integers(100, 100000))).astype(np.uint8)
        elif kind == 2: data = (rng.pareto(0.7, int(rng.integers(100, 130000))) * 3).clip(0, 255).astype(np.uint8)
        elif kind == 3: data = (rng.geometric(0.02 + rng.random() * 0.3, int(rng.integers(100, 130000))) % 256).astype(np.uint8)
        else: data = np.repeat(np.arange(200, dtype=np.uint8), 300)[: int(rng.integers(1000, 60000))]      # many equal counts: the sort's tie order
        hist = np.bincount(data, minlength=256).astype(np.uint32)
        nz = int(np.count_nonzero(hist))
        max_bits = int(rng.integers(7, 12))                                         # low limits force the height limiter
        if nz < 2 or (1 << max_bits) < nz: continue
        max_sym = int(np.nonzero(hist)[0].max())
        outs = []
        for lib, fn in ((e, "emu_huf_build"), (o, "zo_test_huf_build")):
            bits = np.zeros(256, dtype=np.uint8); code = np.zeros(256, dtype=np.uint16)
            lg = getattr(lib, fn)(hist.ctypes.data_as(C.c_void_p), C.c_uint32(max_sym), C.c_uint32(max_bits), bits.ctypes.data_as(C.c_void_p), code.ctypes.data_as(C.c_void_p))
            outs.append((lg, bits.tolist(), code.tolist()))
        assert outs[0] == outs[1], (it, kind, max_sym, max_bits)
        done += 1
    assert done > 300


def test_emulated_decode_pipeline_on_varied_frames(emu, corpus):
    """K1 -> KB -> K1b -> K2 -> K3 on libzstd frames that reach the staging paths of K3: literal runs and far matches above and below
    32 bytes, near matches that start before their batch, overlapped matches (offset < length), raw and RLE blocks, checksums"""
    import numpy as np
    from tests import reflib
    if not reflib.have_ref():
        pytest.skip("needs oracle/_ref (libzstd 1.5.7) to produce the frames")
    ref = reflib.RefZstd()
    rng = np.random.default_rng(5)
    raws = []
    for i in range(20):
        kind = i % 8
        n = int(rng.integers(1000, 131073))
        if kind in (0, 1, 2): r = corpus.frame_bytes(int(rng.integers(0, 2000)))[:n]
        elif kind == 3: r = rng.bytes(n)
        elif kind == 4:
            blk = rng.bytes(700); r = ((blk + rng.bytes(3000) + blk * 5 + rng.bytes(100) + blk) * 20)[:n]
        elif kind == 5:
            a = bytearray(corpus.frame_bytes(int(rng.integers(0, 2000)))[:n])
            for k in range(0, len(a), 97): a[k] = int(rng.integers(0, 256))
            r = bytes(a)
        elif kind == 6:
            parts, tot = [], 0
            while tot < n:
                m = int(rng.integers(1, 3000)); parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0, 256))]) * m); tot += m
            r = b"".join(parts)[:n]
        else: r = bytes(rng.integers(0, 5, n, dtype=np.uint8))
        raws.append(r)
    raws += [b"ab" * 40000, b"x" * 100000, b"0123456789" * 9000]
    # short self-overlapping matches (offset < length <= 32: K3's doubling copy, capped and uncapped): periods 1..9, runs of 5..60 bytes
    for _ in range(2):
        parts = []
        for k in range(3000):
            per = int(rng.integers(1, 10)); run = int(rng.integers(5, 61))
            parts.append((rng.bytes(per) * (run // per + 2))[:run]); parts.append(rng.bytes(int(rng.integers(1, 6))))
        raws.append(b"".join(parts)[:131072])
    frames = [ref.compress(r, level=3, flags=7 if i % 2 else 5) for i, r in enumerate(raws)]
    outs, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in raws], n_blocks=3, chunk=0)
    assert not any(st) and nfb == 0
    assert outs == raws


def test_compact_decode_arenas_run_out_gracefully(emu, oracle, corpus):
    """Round 5: frames take their literal / sequence room from ONE per-chunk arena with a budget (ZhipPipeArgs.bases: K1 claims the literals' room,
    K2 a group's sequences' with one atomic add). A chunk whose budget is used up hands the frames that found no room to the generic kernel:
    every frame still decodes to its bytes, some of them through the fallback list; with room for everybody none does."""
    import ctypes
    import numpy as np
    rng = np.random.default_rng(23)
    raws = [corpus.frame_bytes(i)[: 2000 + 9000 * i] for i in range(12)] + [rng.bytes(30000), b"ab" * 30000, bytes(rng.integers(0, 4, 60000, dtype=np.uint8))]
    frames = [oracle.compress(r, level=3, flags=5) for r in raws]
    sizes = [len(r) for r in raws]
    emu.lib.emu_set_arena_budget.argtypes = [ctypes.c_uint64]
    try:
        for units16, expect_fallback in ((0, False), (20000, True), (1, True)):                  # (units of 16 bytes: room for everybody / for some / for nobody)
            emu.lib.emu_set_arena_budget(units16)
            dec, st, nfb = emu.decompress_pipeline(frames, sizes, n_blocks=3, chunk=0)
            assert not any(st) and dec == raws, units16
            assert (nfb > 0) == expect_fallback, (units16, nfb)
        # the several-block mode claims per ITEM (block): frames of one to three blocks, room for everybody / for some
        big = [corpus.frame_bytes(40 + i) + corpus.frame_bytes(80 + i)[: 70000 * (i % 3)] for i in range(4)]
        bframes = [oracle.compress(r, level=3, flags=5) for r in big]
        emu.set_blocks(4)
        for units16, expect_fallback in ((0, False), (16000, True)):
            emu.lib.emu_set_arena_budget(units16)
            dec, st, nfb = emu.decompress_pipeline(bframes, [len(r) for r in big], n_blocks=3, chunk=0)
            assert not any(st) and dec == big, units16
            assert (nfb > 0) == expect_fallback, (units16, nfb)
    finally:
        emu.lib.emu_set_arena_budget(0)
        emu.set_blocks(0)


def test_emulated_decode_pipeline_with_dictionaries(emu, ref, corpus):
    """dictionary frames through K1 -> KB -> K1b -> K2 -> K3 (r02v; before, every one went to the generic kernel, which rebuilt the
    dictionary's tables per frame): treeless literals and "repeat" sequence tables take the dictionary's ready-made tables, the repeat
    offsets start from the dictionary's, match sources below the frame's first byte come from its content -- wholly, or straddling
    the boundary. Trained and raw-content dictionaries, frames that ignore the dictionary, wrong / missing dictionary."""
    import numpy as np
    rng = np.random.default_rng(7)
    docs = [f[j * 4096:(j + 1) * 4096] for f in corpus.frame_list(700, 24) for j in range(4)]
    trained = ref.train_dictionary(16384, [f[:3000] for f in corpus.frame_list(900, 400)])
    rawd = corpus.frame_bytes(600)[:6000]
    try:
        for dd in (trained, rawd):
            content = dd[-4000:]
            b30 = rng.bytes(30)
            straddlers = [b30 + content[-20:] + b30[:15] + rng.bytes(10) + content[-40:] + b30[:25],      # sources that start in the dictionary and end in the frame
                          content[-300:] + content[-300:] + rng.bytes(5) + content[-64:] + content[-300:-250]]
            raws = docs + [b"", b"a", corpus.frame_bytes(5)[:40000], rawd[1000:5000] + rng.bytes(50) + rawd[:800], (dd[-3000:] + docs[3])[:6000]] + straddlers
            for level in (3, 1):
                frames = [ref.compress(r, level=level, dict_data=dd) for r in raws]
                frames += [ref.compress(docs[0], level=level), ref.compress(rng.bytes(700), level=level, dict_data=dd)]      # a frame made without it; raw block
                want = raws + [docs[0], None]
                assert emu.set_ddict(dd) == 0
                for chunk in (0, 7):
                    outs, st, nfb = emu.decompress_pipeline(frames, [len(r) if r is not None else 700 for r in want], n_blocks=3, chunk=chunk)
                    assert not any(st) and nfb == 0
                    assert all(w is None or o == w for o, w in zip(outs, want)), (level, chunk)
        # wrong dictionary (another id), no dictionary at all: dictionary_wrong (32) for frames that name one, the others still decode
        other = ref.train_dictionary(8192, [f[:2000] for f in corpus.frame_list(300, 300)])
        frames = [ref.compress(docs[1], level=3, dict_data=trained), ref.compress(docs[2], level=3)]
        for dd in (other, None):
            assert emu.set_ddict(dd) == 0
            outs, st, nfb = emu.decompress_pipeline(frames, [4096, 4096], n_blocks=2, chunk=0)
            assert st == [32, 0] and outs[1] == docs[2]
    finally:
        emu.set_ddict(None)


def test_several_block_search_with_an_understated_size_hint(emu, oracle, corpus):
    """the flat search's per-source sequence slices are sized from the caller's size HINT; a source larger than the hint would write
    past its slice. ze_split_body leaves such a source to the generic kernel: frames stay bit-exact, nothing is written out of bounds (the
    ASan build of the emulator runs this too)."""
    import numpy as np
    rng = np.random.default_rng(3)
    raws = [corpus.frame_bytes(i) + corpus.frame_bytes(i + 50)[: 20000 + 30000 * i] for i in range(4)]            # ~150 - 240 KiB
    raws += [corpus.frame_bytes(7) + corpus.frame_bytes(8) + corpus.frame_bytes(9)[:50000], corpus.frame_bytes(11) * 3]          # ~310 / 384 KiB: beyond the hint
    want = [oracle.compress(r, level=3, flags=7) for r in raws]
    s0 = emu.stat(8)
    try:
        emu.set_mb_hint(200 << 10)
        outs, st = emu.compress_batch(raws, level=3, flags=7, n_blocks=2, pipeline=True, chunk=0)
        assert st == [0] * len(raws) and outs == want
    finally:
        emu.set_mb_hint(0)
    assert emu.stat(8) - s0 >= 2                 # the sources within the hint were still searched by the flat kernel


@pytest.mark.parametrize("defines", [["-DZP_HUF_FRAMES=16"], ["-DZP_HUF_FRAMES=8"], ["-DZP_HUF_FRAMES=4", "-DZQ_FRAMES=9", "-DZQ_FENCES=2", "-DZP_ASM_BYTES=2048"]])
def test_decode_shape_variants_stay_correct(oracle, corpus, tmp_path, defines):
    """the build-time shapes decode the same bytes: the quad K2 with fewer frames per wave and other fence placement, K1b with 16 / 4 frames per
    wave, K3 with a smaller assembly buffer (the losing forms of rounds 1-4 -- lane-per-frame K2, K3 / K1b rewrites -- left the source in round 5:
    git tag r04-experiments)"""
    import numpy as np
    from tests import emulib
    emu = emulib.Emu(emulib.build_variant(str(tmp_path / "libzhip_emu_shape.so"), defines))
    rng = np.random.default_rng(5)
    raws = [corpus.frame_bytes(i)[: 3000 + 9000 * i] for i in range(12)] + [rng.bytes(40000), b"ab" * 30000, bytes(rng.integers(0, 6, 70000, dtype=np.uint8))]
    frames = [oracle.compress(r, level=3, flags=7) for r in raws]
    dec, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in raws], n_blocks=3, chunk=0)
    assert not any(st) and dec == raws


def test_explicit_parameters_and_magicless_bit_exact(emu, ref, corpus):
    """ZstdCompressionParameters' fields reach the kernels as per-size-class rows (zhip_cparams.hpp -> ze_get_cparams): frames equal
    libzstd's with the same explicit parameters; what is not implemented (a window smaller than a one-block source, strategies above
    double-fast) is refused per frame with "Unsupported parameter", never encoded differently. Magicless frames both ways."""
    raws = [corpus.frame_bytes(i)[:n] for i, n in [(0, 131072), (1, 50000), (2, 16384), (3, 9000), (9, 300), (10, 20000)]]
    cases = [dict(hash_log=10, chain_log=8), dict(min_match=7), dict(min_match=4, hash_log=17, chain_log=16), dict(strategy=1, target_length=5),
             dict(strategy=2), dict(min_match=3), dict(window_log=12), dict(window_log=17, hash_log=16, chain_log=14, min_match=6, strategy=2)]
    try:
        for level, kw in [(3, c) for c in cases] + [(1, cases[0]), (-3, cases[4]), (19, cases[7])]:
            want = [ref.compress_advanced(r, level=level, **kw) for r in raws]
            emu.set_cparams(**kw)
            outs, st = emu.compress_batch(raws, level=level, pipeline=True)
            for i, (o, w) in enumerate(zip(outs, want)):
                if "window_log" in kw and (1 << kw["window_log"]) < len(raws[i]):
                    assert st[i] == 40, (level, kw, i)                                   # ZSTD_error_parameter_unsupported
                else:
                    assert st[i] == 0 and o == w, (level, kw, i)
        # a window of exactly one block over several full blocks: every later block starts with an EMPTY prefix and skips its first
        # position like the frame's first block does (zstd.c:31091 / :31958; tests/stress_emu_params.py seed 740 found the difference)
        text = b"".join(corpus.frame_list(40, 4))
        blk = corpus.frame_bytes(3)[:700] + corpus.frame_bytes(4)[:2076]
        several = [text[:300000], (blk * 120)[:262144], text[:262144], (blk * 200)[:400000]]
        for strat in (1, 2):
            emu.set_cparams(window_log=17, strategy=strat)
            outs, st = emu.compress_batch(several, level=3, flags=1, pipeline=True)
            assert not any(st) and outs == [ref.compress_advanced(r, level=3, flags=1, window_log=17, strategy=strat) for r in several], strat
        emu.set_cparams(strategy=3)
        assert set(emu.compress_batch(raws[:2], level=3, pipeline=True)[1]) == {40}
        emu.set_cparams(magicless=True)
        want = [ref.compress_advanced(r, level=3, format=1) for r in raws]
        outs, st = emu.compress_batch(raws, level=3, pipeline=True)
        assert st == [0] * len(raws) and outs == want and all(w[:4] != b"\x28\xb5\x2f\xfd" for w in want)
        back, st, nfb = emu.decompress_pipeline(want, [len(r) for r in raws])
        assert st == [0] * len(raws) and back == raws
        back, st = emu.decompress_batch(want, [len(r) for r in raws])
        assert st == [0] * len(raws) and back == raws
    finally:
        emu.set_cparams()


def test_fast_strategy_dictionary_bit_exact(emu, ref, corpus):
    """levels whose row is ZSTD_fast with an attached dictionary (ZSTD_compressBlock_fast_dictMatchState_generic, zstd.c:32197; the
    dictionary's single tagged table from ZSTD_fillHashTableForCDict): trained, raw-content and repetitive dictionaries, sources up to
    the 8 KiB attach cutoff of that strategy; larger sources are refused per frame, never encoded differently"""
    import numpy as np
    rng = np.random.default_rng(8)
    samples = []
    for i in range(128):
        samples += [b"foo" * 64, b"bar" * 64, b"foobar" * 64]
    dicts = [ref.train_dictionary(8192, samples), corpus.frame_bytes(600)[:6000],
             open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict_json4k_16k.bin"), "rb").read()]
    raws = [corpus.frame_bytes(i)[: int(rng.integers(40, 8193))] for i in range(24)] + [b"foo bar foobar foo bar foobar", b"", b"x", b"abcdefgh", b"foobar" * 900]
    for dd in dicts:
        for level in (1, -3):
            want = [ref.compress(r, level=level, dict_data=dd) for r in raws]
            for pipe in (True, False):
                outs, st = emu.compress_batch(raws, level=level, flags=5, pipeline=pipe, dict_data=dd)
                assert not any(st) and outs == want, (level, pipe)


def test_dictionary_table_copy_mode_bit_exact(emu, ref, corpus):
    """sources above libzstd's attach cutoffs (8 KiB fast / 16 KiB double-fast, zstd.c:25250) up to one block: ZSTD_resetCCtx_byCopyingCDict +
    ZSTD_compressBlock_{fast,doubleFast}_extDict_generic (zstd.c:25356, :32423, :31544) -- the dictionary's tables copied with their tags
    stripped, its content an external segment of the window. What three of the reference's own tests do (b"foobar" * 16384 at level 1)."""
    import numpy as np
    rng = np.random.default_rng(9)
    samples = []
    for i in range(128):
        samples += [b"foo" * 64, b"bar" * 64, b"foobar" * 64, b"qwert" * 64, b"yuiop" * 64]
    dicts = [ref.train_dictionary(8192, samples), corpus.frame_bytes(600)[:6000]]
    raws = [b"foobar" * 16384, corpus.frame_bytes(3)[:8193], corpus.frame_bytes(4)[:16385], corpus.frame_bytes(5)[:40000], rng.bytes(20000),
            corpus.frame_bytes(600)[3000:50000], corpus.frame_bytes(6)]
    for dd in dicts:
        for level in (1, 3):
            want = [ref.compress(r, level=level, dict_data=dd) for r in raws]
            for pipe in (True, False):
                outs, st = emu.compress_batch(raws, level=level, flags=5, pipeline=pipe, dict_data=dd)
                assert not any(st) and outs == want, (level, pipe)


def test_dictionary_multi_block_frames_bit_exact(emu, ref, corpus):
    """sources above 128 KiB against a dictionary (ZSTD_compress_frameChunk, zstd.c:27545, over the table-copy state): the dictionary stays
    an external segment for every block, repcodes and the Huffman table carry over, the dictionary's LL / ML tables stay repeatable while
    each block repeats them, its offset table only for the first block (zstd.c:27392). The reference's generate_samples() reaches 196 608 bytes."""
    import numpy as np
    rng = np.random.default_rng(4)
    inputs = [b"foo" * 32, b"bar" * 16, b"abcdef" * 64, b"sometext" * 128, b"baz" * 512]
    samples = []
    for i in range(128):
        samples += [inputs[i % 5], inputs[i % 5] * (i + 3), inputs[-(i % 5)] * (i + 2)]
    big = b"".join(corpus.frame_bytes(i) for i in range(3))
    dicts = [ref.train_dictionary(8192, samples), corpus.frame_bytes(600)[:6000]]
    raws = [samples[-1], samples[-5], big[:131073], big[:200000], b"a" * 300000, rng.bytes(140000), (corpus.frame_bytes(9) + b"x" * 200000)[:250000]]
    for dd in dicts:
        for level in (1, 3):
            want = [ref.compress(r, level=level, dict_data=dd) for r in raws]
            outs, st = emu.compress_batch(raws, level=level, flags=5, pipeline=True, dict_data=dd)
            assert not any(st) and outs == want, level
    # the dictionary stays valid while the source fits the window (level 1: 512 KiB); one byte more and libzstd would drop it part-way: refused
    edge = (big * 2)[:1 << 19]
    outs, st = emu.compress_batch([edge, edge + b"!"], level=1, flags=5, pipeline=True, dict_data=dicts[1])
    assert st == [0, 40] and outs[0] == ref.compress(edge, level=1, dict_data=dicts[1])


def test_damaged_frames_through_the_emulated_pipeline(emu, ref, corpus):
    """libzstd frames with bits flipped, bytes overwritten, pieces cut out or the tail dropped, whole frames as neighbours: what the
    pipeline accepts is byte-for-byte what libzstd (zstd/zstd.c:44174 ZSTD_decompressFrame) makes of the same bytes, nothing libzstd
    rejects gets through, the neighbours decode. The other direction is allowed to differ in one documented way (DESIGN.md section 2:
    damaged Huffman streams libzstd's fast loop lets through are refused) and is bounded here. tests/stress_emu_corrupt.py is the
    open-ended form, run under AddressSanitizer (tests/emu/build_asan.sh) for the bounds."""
    import numpy as np
    from tests.stress_emu_corrupt import one_round
    tot = {}
    rng = np.random.default_rng(11)
    for k in range(3):
        r = one_round(emu, ref, rng, corpus, count=24, small=(k != 0), big=(k == 2))
        for a, b in r.items(): tot[a] = tot.get(a, 0) + b
    trained = ref.train_dictionary(16384, [f[j * 4096:(j + 1) * 4096] for f in corpus.frame_list(900, 24) for j in range(16)])
    try:
        assert emu.set_ddict(trained) == 0
        r = one_round(emu, ref, rng, corpus, count=24, small=True, dict_data=trained)
        for a, b in r.items(): tot[a] = tot.get(a, 0) + b
    finally:
        emu.set_ddict(None)
    assert tot["wrong"] == 0 and tot["missed"] == 0 and tot["neighbours_bad"] == 0, tot
    assert tot["rejected"] >= 30 and tot["accepted"] >= 24 and tot["stricter"] <= tot["frames"] // 10, tot


def test_precomputed_dictionary_parameters_through_the_kernels(emu, ref, corpus):
    """How ZstdCompressionDict.precompute_compress reaches the kernels (cext/backend_hip.c dict_precompute / apply_precomputed): level 3 plus
    the precomputed dictionary's six non-window fields as explicit parameters, the frame's window the compressor's own. Against libzstd
    driven the way the reference drives it (ZSTD_createCDict_advanced + ZSTD_CCtx_refCDict, c-ext/compressiondict.c:266-278,
    compressor.c:29-31): the dictionary's level wins over the compressor's, in attach mode, table-copy mode and over several blocks."""
    import numpy as np
    pool = corpus.frame_list(0, 6)
    rng = np.random.default_rng(3)
    dicts = [ref.train_dictionary(16384, [f[j * 4096:(j + 1) * 4096] for f in pool for j in range(16)]), pool[3][1000:9000]]
    differs = 0
    try:
        for dd in dicts:
            raws = [(pool[0] + pool[1] + pool[2])[:n] for n in (1, 300, 4096, 8193, 16385, 60000, 131073, 200000)] + [rng.bytes(3000), (dd[-3000:] + pool[5])[:30000]]
            for plevel, clevel, window in ((1, 3, 0), (3, 1, 0), (-3, 3, 0), (1, 3, 19)):
                pre = ref.cdict_params(plevel, len(dd))
                pre["window_log"] = window                      # 0: the default level's row, else the compressor's explicit window
                want = [ref.compress_with_cdict(r, dd, level=clevel, cdict_level=plevel, window_log=window) for r in raws]
                differs += sum(1 for r, w in zip(raws, want) if w != ref.compress(r, level=clevel, dict_data=dd))
                emu.set_cparams(**pre)
                for pipe in (True, False):
                    outs, st = emu.compress_batch(raws, level=3, flags=5, n_blocks=2, pipeline=pipe, dict_data=dd)
                    assert not any(st) and outs == want, (len(dd), plevel, clevel, window, pipe)
    finally:
        emu.set_cparams()
    assert differs > 20          # the precomputed level really changes the frames


def test_flat_dictionary_search_bit_exact(emu, ref, corpus):
    """ze_dfast_dict_flat (round 3): the attached-dictionary double-fast search in the flat kernel's form -- one lane per document, unconditional
    load rounds, tables zeroed by the wave -- against libzstd for the configs[3] dictionary (112 640 bytes: 384 KiB of tagged tables, 110 KiB of
    content), the 16 KiB one, and raw-content dictionaries; documents, sources that straddle into / quote from deep inside the content, both
    sides of the 16 KiB attach cutoff, tiny and empty sources (ZSTD_compressBlock_doubleFast_dictMatchState_generic, zstd.c:31262)"""
    import numpy as np
    from tests.corpus import Corpus
    rng = np.random.default_rng(31)
    here = os.path.dirname(os.path.abspath(__file__))
    big = open(os.path.join(here, "golden", "dict_json4k.bin"), "rb").read()
    small = open(os.path.join(here, "golden", "dict_json4k_16k.bin"), "rb").read()
    docs = Corpus(frame_size=4096).json_docs(0, 40).numpy()
    raws = [docs[i].tobytes() for i in range(40)]
    raws += [big[-3000:] + raws[0][:1000], big[5000:6500] + raws[1][:2000] + big[60000:61000], big[-100:], b"", b"x", big[-16384:], big[-16385:] + b"!",
             (big[-700:] * 30)[:16384], raws[5] * 4, raws[6] * 3 + b"zz"]
    raws += [corpus.frame_bytes(i)[: int(rng.integers(40, 16385))] for i in range(8)]
    raws += [bytes(rng.integers(0, 4, int(rng.integers(64, 9000)), dtype=np.uint8)) for _ in range(3)]
    for dd in (big, small, corpus.frame_bytes(600)[:6000], corpus.frame_bytes(601)[:100000]):
        want = [ref.compress(r, level=3, dict_data=dd) for r in raws]
        before = emu.lib.emu_stat(15)
        outs, st = emu.compress_batch(raws, level=3, flags=5, pipeline=True, dict_data=dd)
        assert not any(st) and outs == want
        assert emu.lib.emu_stat(15) - before >= 50          # the flat kernel searched them (sources under 64 bytes go to the lane-serial kernel)
    # round 4: table / arena slots sized from the caller's size hint (ZhipEncodeArgs.slotSrcMax): the match kernels take sources up to 4 KiB,
    # the rest -- also sources BELOW the attach cutoff -- goes to the generic kernel; the frames are libzstd's either way
    try:
        emu.set_dict_slot_max(4096)
        want = [ref.compress(r, level=3, dict_data=big) for r in raws]
        before = emu.lib.emu_stat(15)
        outs, st = emu.compress_batch(raws, level=3, flags=5, pipeline=True, dict_data=big)
        assert not any(st) and outs == want
        assert 40 <= emu.lib.emu_stat(15) - before < 50
    finally:
        emu.set_dict_slot_max(0)


def test_decode_pipeline_long_items_and_carried_flush(emu, ref):
    """K3's paths for items above its own-lane threshold (16-byte units dealt out to the lanes), overlapping matches, and the flush that carries a
    batch's last < 16 bytes over to the next batch: frames made of long literal runs, long far matches, byte runs and short pieces, levels 1 - 19"""
    import numpy as np
    rng = np.random.default_rng(5)
    raws = []
    for t in range(6):
        parts = []; pool = [rng.bytes(int(rng.integers(40, 5000))) for _ in range(6)]
        while sum(map(len, parts)) < 120000:
            k = int(rng.integers(0, 4))
            parts.append(rng.bytes(int(rng.integers(1, 3000))) if k == 0 else pool[int(rng.integers(0, 6))] if k == 1
                         else bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 2500)) if k == 2 else pool[int(rng.integers(0, 6))][: int(rng.integers(1, 60))])
        raws.append(b"".join(parts)[: int(rng.integers(100000, 131073))])
    for lv in (1, 3, 19):
        frames = [ref.compress(r, level=lv) for r in raws]
        outs, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in raws])
        assert not any(st) and outs == raws and nfb == 0


def test_frames_no_encoder_writes_through_the_emulated_kernels(emu, ref):
    """tests/craft.py's hand-made frames (repeat offset 1 minus one = 0, zstd.c:46941; blocks above the frame's block maximum under libzstd's
    one-pass and streaming decoders, zstd.c:44239-44246 / :47714): K1 / K2 / K3 and the generic kernel accept exactly what libzstd accepts and
    produce its bytes. The GPU form is tests/test_gpu_decompress.py::test_frames_no_encoder_writes_are_answered_like_libzstd."""
    from tests import craft
    cases = craft.edge_frames() + craft.skippable_frames() + craft.encoding_variants()
    outs, st, nfb = emu.decompress_pipeline([c[1] for c in cases], [c[2] for c in cases], n_blocks=3, chunk=0)
    outs2, st2 = emu.decompress_batch([c[1] for c in cases], [c[2] for c in cases], n_blocks=2)          # every one through the generic kernel too
    for (name, f, n, ok), o, s, o2, s2 in zip(cases, outs, st, outs2, st2):
        try: want = ref.decompress(f, n)
        except RuntimeError: want = None
        assert (want is not None) == ok, name
        assert (s == 0) == ok and (not ok or o == want), (name, s)
        assert (s2 == 0) == ok and (not ok or o2 == want), (name, s2, "generic kernel")


def test_a_whole_block_as_one_match_into_the_dictionary(emu, ref):
    """match length 131 072 (a 128 KiB source that is its own raw-content dictionary): 18 bits in the packed sequences K2 writes for K3.
    GPU form: tests/test_gpu_boundary.py, same name."""
    import numpy as np
    data = np.random.default_rng(5).bytes(131072)
    blob = b"zz" + data
    raws = [data, data[:131071], data[:3] + b"Q" + data[:131068]]
    frames = [ref.compress(r, level=3, dict_data=blob) for r in raws]
    assert len(frames[0]) < 40
    try:
        assert emu.set_ddict(blob, raw_content=True) == 0
        outs, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in raws], n_blocks=2, chunk=0)
    finally:
        emu.set_ddict(None)
    assert st == [0, 0, 0] and nfb == 0 and outs == raws
    outs, st = emu.compress_batch(raws[:1], level=3, flags=5, pipeline=True, dict_data=blob)
    assert st == [0] and outs[0] == frames[0]


def test_frames_of_several_blocks_through_the_pipeline(emu, ref, corpus):
    """The decode pipeline's several-block mode (zhip_format.hpp ZpFrameRec: K1 per frame over its blocks, K1b / K2 per block, K3 per frame
    over its blocks, the repeat-offset history symbolic across block boundaries): frames of 2-4 blocks of libzstd (ZSTD_compress_frameChunk,
    zstd/zstd.c:27545 -- pre-split blocks, raw and RLE blocks, 'treeless' and 'repeat' entropy tables, checksums) decode to their sources;
    with too few item slots the overflowing frames are the generic kernel's; tests/craft.py's hand-made frames get libzstd's answers in this
    mode too; damaged frames: nothing wrong gets through. tests/stress_emu_decode_any.py / stress_emu_corrupt.py (ZHIP_EMU_BLOCKS) go on."""
    import numpy as np
    from tests import craft
    from tests.stress_emu_corrupt import one_round
    rng = np.random.default_rng(21)
    raws = [b"".join(corpus.frame_bytes(300 + 4 * i + k) for k in range(4))[: int(rng.integers(140000, 400000))] for i in range(3)]
    raws += [corpus.frame_bytes(7)[:50000], rng.bytes(280000), b"\x07" * 300000, (corpus.frame_bytes(9)[:700] + rng.bytes(90)) * 400]
    frames = [ref.compress(r, level=[3, 1, 3, 5, 3, 3, 3][i], flags=7 if i % 2 else 5) for i, r in enumerate(raws)]
    sizes = [len(r) for r in raws]
    try:
        for per_frame, chunk in ((8, 0), (2, 3)):
            emu.set_blocks(per_frame)
            outs, st, nfb = emu.decompress_pipeline(frames, sizes, n_blocks=3, chunk=chunk)
            assert st == [0] * len(raws) and outs == raws, (per_frame, st)
            assert (nfb == 0) if per_frame == 8 else (nfb >= 3), (per_frame, nfb)
        emu.set_blocks(6)
        cases = craft.edge_frames()
        outs, st, nfb = emu.decompress_pipeline([c[1] for c in cases], [c[2] for c in cases], n_blocks=3, chunk=0)
        for (name, f, n, ok), o, s in zip(cases, outs, st):
            assert (s == 0) == ok and (not ok or o == ref.decompress(f, n)), (name, s)
        r = one_round(emu, ref, rng, corpus, count=24, small=False, big=True)
        assert r["wrong"] == 0 and r["missed"] == 0 and r["neighbours_bad"] == 0 and r["accepted"] >= 6, r
        dd = corpus.frame_bytes(950)[:60000]                                          # a raw-content dictionary: matches reach below the frame's first byte from any block
        draws = [raws[0], dd[200:9000] * 20]
        dframes = [ref.compress(x, level=3, flags=7, dict_data=dd) for x in draws]
        assert emu.set_ddict(dd, raw_content=True) == 0
        outs, st, nfb = emu.decompress_pipeline(dframes, [len(x) for x in draws], n_blocks=2, chunk=0)
        assert st == [0, 0] and outs == draws and nfb == 0
    finally:
        emu.set_blocks(0); emu.set_ddict(None)


def test_sources_of_several_blocks_in_the_flat_search(emu, ref, corpus):
    """The compress pipeline on sources above 128 KiB with the flat match kernel searching them (ZeMbBlock, zhip_format.hpp; the split kernel's
    block layout, the generic kernel's entropy coding + check of the search's assumption + redo): bit-exact against libzstd
    (ZSTD_compress_frameChunk, zstd/zstd.c:27545) on sources built to break the assumption now and then; too few block slots send a source to
    the generic kernel's own search. tests/stress_emu_encode_blocks.py is the open-ended form; GPU: tests/test_gpu_compress.py, same name."""
    import numpy as np
    from tests.stress_emu_encode_blocks import make
    rng = np.random.default_rng(8)
    raws = [make(rng, corpus) for _ in range(10)] + [corpus.frame_bytes(3)[:70000]]
    want = [ref.compress(r, level=3, flags=7) for r in raws]
    s0, r0 = emu.stat(8), emu.stat(9)
    try:
        for slots, chunk, probes in ((1, 0, 2), (3, 4, 2), (1, 0, 4)):          # (four probes per trip: the form chunks of up to 32 768 block-sized pieces take)
            emu.set_mb_compress(slots); emu.lib.emu_set_probes(probes)
            outs, st = emu.compress_batch(raws, level=3, flags=7, n_blocks=2, pipeline=True, chunk=chunk)
            assert st == [0] * len(raws) and outs == want, (slots, probes, st)
    finally:
        emu.set_mb_compress(1); emu.lib.emu_set_probes(2)
    assert emu.stat(8) - s0 >= 8 and emu.stat(9) - r0 >= 1          # most sources searched by the flat kernel, at least one redone


def test_k1_lane_per_frame_for_dictionary_batches(emu, ref, corpus):
    """Round 6: in dictionary batches K1's waves take several frames at a time and a LANE walks a frame whose tables are all the
    dictionary's (zp_lit_shared_try: treeless / raw / RLE literals, every sequence table "repeat"); what a lane cannot finish -- a table of the frame's own,
    another layout, any failing check -- is listed for K1 proper, which does it as before. The wave-per-frame form alone (k1Lanes = 0) and the two-pass form
    must agree on every output byte and every status, on good frames, on frames that need the wave, and on damaged ones; and the lanes must in fact finish
    the 4 KiB documents of BASELINE configs[3]'s shape."""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(23)
    docs = [f[j * 4096:(j + 1) * 4096] for f in corpus.frame_list(700, 20) for j in range(8)]            # 160 documents
    trained = ref.train_dictionary(16384, [f[j * 4096:(j + 1) * 4096] for f in corpus.frame_list(900, 24) for j in range(16)])
    raws = docs + [b"", b"a", corpus.frame_bytes(5)[:40000], corpus.frame_bytes(9), rng.bytes(900), bytes(3000), docs[3][:100] * 30]
    frames = [ref.compress(r, level=3, dict_data=trained) for r in raws]
    frames += [ref.compress(docs[0], level=3), ref.compress(docs[1], level=1, dict_data=trained), ref.compress(docs[2], level=3, flags=7, dict_data=trained)]
    sizes = [len(r) for r in raws] + [4096, 4096, 4096]
    # damaged copies: a bit in the literals header, the sequences header, the streams, the tail cut
    bad = []
    for k in range(40):
        f = bytearray(frames[k])
        where = [6, 9, 10, 11, 12, len(f) // 2, len(f) - 2][k % 7]
        if k % 8 == 7:
            f = f[:len(f) - 1 - k % 5]
        else:
            f[min(where, len(f) - 1)] ^= 1 << (k % 8)
        bad.append(bytes(f))
    allf = frames + bad
    alls = sizes + [4096] * len(bad)
    try:
        assert emu.set_ddict(trained) == 0
        res = {}
        for lanes in (0, 1):
            emu.lib.emu_set_k1_lanes(C.c_uint32(lanes))
            before = emu.stat(7)
            outs, st, nfb = emu.decompress_pipeline(allf, alls, n_blocks=3, chunk=0)
            res[lanes] = (outs, st, nfb, emu.stat(7) - before)
        for lanes in (1,):
            assert res[lanes][1] == res[0][1], (lanes, [(i, a, b) for i, (a, b) in enumerate(zip(res[lanes][1], res[0][1])) if a != b][:5])
            assert all(a == b for a, b, s in zip(res[lanes][0], res[0][0], res[0][1]) if s == 0), lanes
            assert res[lanes][2] == res[0][2]
        outs, st = res[1][0], res[1][1]
        assert not any(st[:len(frames)]) and all(o == r for o, r in zip(outs, raws + [docs[0], docs[1], docs[2]]))
        assert res[0][3] == 0 and res[1][3] >= 30, (res[0][3], res[1][3])              # (corpus slices often carry tables of their own)
        assert any(st[len(frames):])                                                                                  # (and the damage was noticed)
        # BASELINE configs[3]'s own shape: JSON-like 4 KiB documents on the 112 640-byte trained dictionary -- every frame is header arithmetic
        import os
        from tests.corpus import Corpus
        jd = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict_json4k.bin"), "rb").read()
        jraw = Corpus(frame_size=4096).json_docs(0, 96).cpu().numpy()
        jdocs = [jraw[i].tobytes() for i in range(96)]
        jframes = [ref.compress(d, level=3, dict_data=jd) for d in jdocs]
        assert emu.set_ddict(jd) == 0
        for lanes in (0, 1):
            emu.lib.emu_set_k1_lanes(C.c_uint32(lanes))
            before = emu.stat(7)
            outs, st, nfb = emu.decompress_pipeline(jframes, [4096] * 96, n_blocks=3, chunk=0)
            assert not any(st) and nfb == 0 and outs == jdocs
            assert emu.stat(7) - before == (96 if lanes else 0)
    finally:
        emu.lib.emu_set_k1_lanes(C.c_uint32(1))
        emu.set_ddict(None)


def test_k0_lane_per_frame_parsers_agree_with_k1s_own(emu, ref, corpus):
    """Round 6: K0 (zp_pre_body) -- a LANE per frame walks what K1's lane 0 used to: the Huffman weights' description (distribution, 64-cell FSE table, two-state
    decode) and the three sequence distributions -- and K1 takes weights and counts from its record. With and without K0 the pipeline must give the same bytes AND the
    same status for every frame: good ones of every class and level, frames whose descriptions are damaged (bit flips confined to the description regions, the tail cut),
    frames with 4-bit weights, RLE / predefined tables, raw / RLE literals; and K1 must in fact have taken the records (test hooks [10] / [11])."""
    import numpy as np
    rng = np.random.default_rng(31)
    raws = []
    for i in range(40):
        base = corpus.frame_bytes(int(rng.integers(0, 4000)))
        n = int(rng.integers(300, 20000)) if i % 5 else 131072
        k = i % 6
        if k == 0: r = base[:n]
        elif k == 1: r = bytes(rng.integers(0, 12, n, dtype=np.uint8))                      # few symbols: 4-bit weights, short tables
        elif k == 2: r = (base[:300] + bytes(rng.integers(97, 105, 80, dtype=np.uint8))) * (n // 380 + 1)
        elif k == 3: r = bytes((np.frombuffer(base[:n], dtype=np.uint8) & 0x3F).tobytes())
        elif k == 4: r = rng.bytes(n // 8) + base[:n]
        else: r = base[:n // 2] + bytes(n // 2)
        raws.append(r[:n])
    levels = [3, 1, 3, 5, 3, 9, 3, -1]
    frames = [ref.compress(r, level=levels[i % len(levels)]) for i, r in enumerate(raws)]
    sizes = [len(r) for r in raws]
    bad, bsz = [], []
    for k in range(120):
        j = k % len(frames)
        if len(raws[j]) > 20000:
            continue
        f = bytearray(frames[j])
        if k % 10 == 9:
            f = f[:max(6, len(f) - 1 - k % 7)]
        else:
            hi = min(len(f), 200)
            for _ in range(1 + k % 2):
                f[int(rng.integers(5, hi))] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(f)); bsz.append(sizes[j])
    allf, alls = frames + bad, sizes + bsz
    try:
        res = {}
        for k0 in (0, 1):
            emu.set_k0(k0)
            b10, b11 = emu.stat(10), emu.stat(11)
            outs, st, nfb = emu.decompress_pipeline(allf, alls, n_blocks=3, chunk=0)
            res[k0] = (outs, st, nfb, emu.stat(10) - b10, emu.stat(11) - b11)
        assert res[1][1] == res[0][1], [(i, a, b) for i, (a, b) in enumerate(zip(res[1][1], res[0][1])) if a != b][:8]
        assert all(a == b for a, b, s in zip(res[1][0], res[0][0], res[0][1]) if s == 0)
        assert res[1][2] == res[0][2]
        assert not any(res[1][1][:len(frames)]) and all(o == r for o, r in zip(res[1][0], raws))
        assert any(res[1][1][len(frames):])                                     # (the damage was noticed)
        assert res[0][3] == 0 and res[0][4] == 0
        assert res[1][3] >= 10 and res[1][4] >= 30, (res[1][3], res[1][4])       # weights of >= 10 frames, >= 30 distributions came from K0's records
    finally:
        emu.set_k0(1)


def test_flat_dictionary_search_with_launch_numbers_in_the_cells(emu, ref, corpus):
    """Round 6: the flat dictionary search's per-document tables are not zeroed per launch any more -- a cell carries its launch's number above the index
    (ZhipEncodeArgs.tabEpoch) and a cell of any other launch reads as empty. Six launches over the SAME persistent tables with the documents shuffled from launch to
    launch (so every slot holds another document's cells from the launch before), two dictionaries in turn (the index width changes: the tables start over): every
    frame libzstd's, every time."""
    import numpy as np
    from tests.corpus import Corpus
    rng = np.random.default_rng(47)
    here = os.path.dirname(os.path.abspath(__file__))
    big = open(os.path.join(here, "golden", "dict_json4k.bin"), "rb").read()
    small = open(os.path.join(here, "golden", "dict_json4k_16k.bin"), "rb").read()
    docs = Corpus(frame_size=4096).json_docs(0, 48).numpy()
    raws = [docs[i].tobytes() for i in range(48)]
    raws += [big[-3000:] + raws[0][:1000], big[5000:6500] + raws[1][:2000] + big[60000:61000], (big[-700:] * 30)[:16384], raws[5] * 4, raws[6] * 3 + b"zz"]
    raws += [corpus.frame_bytes(i)[: int(rng.integers(100, 16385))] for i in range(6)]
    want = {id(dd): {r: ref.compress(r, level=3, dict_data=dd) for r in set(raws)} for dd in (big, small)}
    try:
        emu.set_dict_epochs(1)
        for launch, dd in enumerate((big, big, big, small, small, big)):
            order = list(rng.permutation(len(raws)))
            batch = [raws[i] for i in order]
            before = emu.lib.emu_stat(15)
            outs, st = emu.compress_batch(batch, level=3, flags=5, pipeline=True, dict_data=dd)
            assert not any(st), launch
            assert outs == [want[id(dd)][r] for r in batch], launch
            assert emu.lib.emu_stat(15) - before >= 50
    finally:
        emu.set_dict_epochs(0)


def test_flat_search_with_launch_numbers_in_the_cells(emu, ref, corpus):
    """The dictionary-less flat search of one-block sources the same way (round 6): cells are position 18 | tag 8 | launch number 6 bits, the tables persist and are zeroed once per 63
    launches. 70 launches over the same persistent tables (so the numbers wrap and the tables start over once), sources shuffled and of changing kinds from launch to launch, two, three
    and four probes per trip and the LDS-source kernel in turn: every frame libzstd's."""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(53)
    pool = [corpus.frame_bytes(200 + i)[: int(rng.integers(2000, 40000))] for i in range(10)]
    pool += [bytes(rng.integers(0, 5, 9000, dtype=np.uint8)), (corpus.frame_bytes(3)[:700] * 40)[:25000], corpus.frame_bytes(77), b"ab" * 3000]
    want = {r: ref.compress(r, level=3) for r in pool}
    try:
        emu.set_dict_epochs(1)
        for launch in range(70):
            emu.lib.emu_set_probes(C.c_uint32((2, 3, 4)[launch % 3]))
            emu.lib.emu_set_e1lds_max(C.c_uint32(64 if launch % 5 == 4 else 0))
            k = 3 if launch not in (0, 33, 69) else len(pool)                      # (most launches small -- the emulator is slow --, three of them the whole pool)
            batch = [pool[i] for i in rng.permutation(len(pool))[:k]]
            outs, st = emu.compress_batch(batch, level=3, flags=5, pipeline=True)
            assert not any(st), launch
            assert outs == [want[r] for r in batch], launch
    finally:
        emu.set_dict_epochs(0); emu.lib.emu_set_probes(C.c_uint32(2)); emu.lib.emu_set_e1lds_max(C.c_uint32(0))


def test_content_checksums_verified_by_a_lane_per_frame(emu, ref, corpus):
    """Round 6: KX (zp_check_body) verifies content checksums after K3, a lane per frame -- K3 and K1 (raw / RLE frames) used to hash 128 KiB on ONE lane. Frames with and without
    checksums side by side: compressible, incompressible (raw block: K1 finishes them), byte runs (RLE), empty, frames of several blocks (the several-block mode), dictionary frames
    that the lane pass finishes -- each also with its checksum trailer or a content byte damaged. Both forms must agree on every status and byte, wrong checksums must be refused as
    checksum_wrong (22), right ones accepted."""
    import numpy as np
    from tests import reflib
    rng = np.random.default_rng(61)
    raws = [corpus.frame_bytes(700 + i)[: int(rng.integers(100, 131073))] for i in range(10)] + [rng.bytes(50000), rng.bytes(131072), b"z" * 70000, b"", b"ab" * 9, bytes(4000)]
    F = reflib.DEFAULT_FLAGS | reflib.F_CHECKSUM
    frames = [ref.compress(r, level=3, flags=F) for r in raws] + [ref.compress(r, level=3) for r in raws[:4]]
    sizes = [len(r) for r in raws] + [len(r) for r in raws[:4]]
    bad = []
    for k in range(len(raws)):
        f = bytearray(frames[k])
        if len(f) < 12: continue
        f[-1 - (k % 4)] ^= 0x40                                  # the trailer itself
        bad.append((bytes(f), sizes[k]))
    allf = frames + [b for b, _ in bad]; alls = sizes + [n for _, n in bad]
    try:
        res = {}
        for later in (0, 1):
            emu.set_check_later(later)
            res[later] = emu.decompress_pipeline(allf, alls, n_blocks=3, chunk=0)
        assert res[1][1] == res[0][1], [(i, a, b) for i, (a, b) in enumerate(zip(res[1][1], res[0][1])) if a != b][:8]
        assert all(a == b for a, b, s in zip(res[1][0], res[0][0], res[0][1]) if s == 0) and res[1][2] == res[0][2]
        st = res[1][1]
        assert not any(st[:len(frames)]) and all(o == r for o, r in zip(res[1][0], raws + raws[:4]))
        assert all(s == 22 for s in st[len(frames):]), st[len(frames):]
        # frames of several blocks
        big = [corpus.frame_bytes(40 + k)[:90000] + rng.bytes(3000) + corpus.frame_bytes(80 + k) for k in range(4)]
        bf = [ref.compress(r, level=3, flags=F) for r in big]
        wrong = bytearray(bf[1]); wrong[-2] ^= 1
        emu.set_blocks(6)
        for later in (0, 1):
            emu.set_check_later(later)
            outs, st, nfb = emu.decompress_pipeline(bf + [bytes(wrong)], [len(r) for r in big] + [len(big[1])], n_blocks=3, chunk=0)
            assert st[:4] == [0, 0, 0, 0] and outs[:4] == big and st[4] == 22, (later, st)
        emu.set_blocks(0)
        # dictionary frames the lane pass finishes, with checksums
        docs = [corpus.frame_bytes(900 + i)[i * 31: i * 31 + 4096] for i in range(70)]
        dd = ref.train_dictionary(16384, [corpus.frame_bytes(950 + i)[:4096] for i in range(120)])
        df = [ref.compress(d, level=3, flags=F, dict_data=dd) for d in docs]
        w2 = bytearray(df[5]); w2[-1] ^= 0x80
        assert emu.set_ddict(dd) == 0
        for later in (0, 1):
            emu.set_check_later(later)
            outs, st, nfb = emu.decompress_pipeline(df + [bytes(w2)], [4096] * 71, n_blocks=3, chunk=0)
            assert not any(st[:70]) and outs[:70] == docs and st[70] == 22, (later, st[60:])
    finally:
        emu.set_check_later(1); emu.set_blocks(0); emu.set_ddict(None)


def test_checksum_trailers_written_by_a_lane_per_frame(emu, ref, corpus):
    """Round 6: EX (ze_trailer_body) fills the frames' checksum trailers after the entropy kernel, a lane per frame (that kernel hashed the source on ONE lane of the frame's wave).
    write_checksum frames of compressible, incompressible (raw block), run (RLE block), tiny and empty sources, with and without a dictionary, with the trailer written either way:
    libzstd's frames, byte for byte."""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(67)
    raws = [corpus.frame_bytes(800 + i)[: int(rng.integers(64, 131073))] for i in range(12)] + [rng.bytes(40000), b"r" * 30000, b"xyz", b"", corpus.frame_bytes(3)[:63]]
    dd = corpus.frame_bytes(990)[:20000]
    docs = [corpus.frame_bytes(990)[i * 50: i * 50 + 3000] + raws[i % 12][:1000] for i in range(20)]
    try:
        for later in (1, 0):
            emu.lib.emu_set_trailer_later(C.c_uint32(later))
            outs, st = emu.compress_batch(raws, level=3, flags=7, pipeline=True)
            assert not any(st) and outs == [ref.compress(r, level=3, flags=7) for r in raws], later
            outs, st = emu.compress_batch(docs, level=3, flags=7, pipeline=True, dict_data=dd)
            assert not any(st) and outs == [ref.compress(r, level=3, flags=7, dict_data=dd) for r in docs], later
    finally:
        emu.lib.emu_set_trailer_later(C.c_uint32(1))
