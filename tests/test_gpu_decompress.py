"""GPU parity tests for the decode path: HIP kernels (through the C ABI) vs the CPU oracle and the originals."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zstd():
    import zstandard_amd
    assert zstandard_amd._lib.lib().zhip_device_count() >= 1, "no GPU visible"
    return zstandard_amd


def _enc():
    from tests import reflib
    return reflib.checker()               # libzstd 1.5.7 itself (reference build, else the image's copy) or an error -- never the restatement


def test_wave_primitives_selftest(zstd):
    """scan (DPP), ballot, shuffles, readfirstlane as the kernels use them"""
    assert zstd._lib.lib().zhip_selftest() == 0, zstd._lib.last_error()


def test_small_and_edge_frames(zstd, oracle):
    enc = _enc()
    raws = [b"", b"foo", b"foo" * 4, b"bar" * 6, b"a" * 1000, b"a" * 131072, bytes(range(256)) * 40,
            b"hello world, hello world, hello there world! " * 500, np.random.default_rng(1).bytes(1 << 17)]
    frames = [enc.compress(r) for r in raws]
    res = zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
    assert len(res) == len(raws)
    assert res.size() == sum(map(len, raws))
    for i, r in enumerate(raws):
        assert res[i].tobytes() == r == oracle.decompress(frames[i], len(r))
    assert res[0].offset == 0


def test_corpus_frames_match_oracle(zstd, oracle, corpus):
    enc = _enc()
    n = 96
    raws = [corpus.frame_bytes(i) for i in range(n)]
    # ragged sizes as well
    raws += [corpus.frame_bytes(100 + i)[: 1000 * (i + 1) + i] for i in range(32)]
    frames = [enc.compress(r) for r in raws]
    res = zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
    for i, r in enumerate(raws):
        got = res[i].tobytes()
        assert got == r, "frame %d" % i
    for i in range(0, len(raws), 7):
        assert res[i].tobytes() == oracle.decompress(frames[i], len(raws[i]))


def test_multi_block_and_levels(zstd):
    from tests import reflib
    if not reflib.have_ref():
        pytest.fail("no libzstd 1.5.7 to check against: " + "needs reference libzstd for other levels")
    ref = reflib.RefZstd()
    from tests.corpus import Corpus
    c = Corpus()
    big = b"".join(c.frame_bytes(i) for i in range(5))       # 640 KiB: 5 blocks, repeat modes, window carry
    for level in (1, 3, 5, 9, 19):
        for flags in (reflib.DEFAULT_FLAGS, reflib.F_DICTID, reflib.DEFAULT_FLAGS | reflib.F_CHECKSUM):
            frame = ref.compress(big, level=level, flags=flags)
            sizes = struct.pack("=Q", len(big))
            res = zstd.ZstdDecompressor().multi_decompress_to_buffer([frame], decompressed_sizes=sizes)
            assert res[0].tobytes() == big, (level, flags)


def test_one_shot_decompress(zstd):
    enc = _enc()
    d = zstd.ZstdDecompressor()
    for raw in (b"foobar" * 256, b"", b"x"):
        assert d.decompress(enc.compress(raw)) == raw
    with pytest.raises(zstd.ZstdError, match="error determining content size from frame header"):
        d.decompress(b"foobar")


def test_item_failure_reports_first_bad_frame(zstd):
    enc = _enc()
    frames = [enc.compress(b"x" * 128), enc.compress(b"y" * 128)]
    frames[1] = frames[1][0:15] + b"extra" + frames[1][15:]
    with pytest.raises(zstd.ZstdError, match="error decompressing item 1: (Data corruption detected|Destination buffer is too small)"):
        zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
    with pytest.raises(ValueError, match="could not determine decompressed size of item 0"):
        zstd.ZstdDecompressor().multi_decompress_to_buffer([b"foobarbaz"])


def test_truncated_and_corrupt_frames_do_not_crash(zstd, corpus):
    enc = _enc()
    raw = corpus.frame_bytes(3)
    frame = enc.compress(raw)
    rng = np.random.default_rng(7)
    bad = []
    for k in range(40):
        b = bytearray(frame)
        if k % 2:
            del b[len(b) // 2 + k:]
        else:
            for _ in range(3):
                b[int(rng.integers(9, len(b)))] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    sizes = struct.pack("=%dQ" % len(bad), *([len(raw)] * len(bad)))
    d = zstd.ZstdDecompressor()
    # the classification DESIGN.md section 2 states, on the GPU: whatever we accept is byte-for-byte what libzstd decodes, and nothing
    # libzstd rejects is accepted (we may refuse a damaged frame libzstd decodes to something: its fast Huffman loop does not check
    # that every stream is consumed exactly, zstd.c:40106-40155 -- never the other way round)
    accepted = refused_both = refused_only_here = 0
    for b in bad:
        try:
            want = enc.decompress(b, len(raw)) if hasattr(enc, "lib") else None
            ref_ok = True
        except RuntimeError:
            want, ref_ok = None, False
        try:
            got = d.multi_decompress_to_buffer([b], decompressed_sizes=sizes[:8])[0].tobytes()
        except zstd.ZstdError:
            got = None
        if got is not None:
            assert ref_ok, "accepted a frame libzstd rejects"
            assert got == want, "accepted a damaged frame but decoded it differently from libzstd"
            accepted += 1
        elif ref_ok:
            refused_only_here += 1
        else:
            refused_both += 1
    assert accepted + refused_both + refused_only_here == len(bad) and refused_both >= 20      # (every truncation is refused by both)
    # the whole batch at once: the damaged frames do not disturb their neighbours
    good = enc.compress(corpus.frame_bytes(4))
    with pytest.raises(zstd.ZstdError, match="error decompressing item 1"):
        d.multi_decompress_to_buffer([good, bad[1], good], decompressed_sizes=sizes[:24])
    assert d.multi_decompress_to_buffer([good, good])[1].tobytes() == corpus.frame_bytes(4)


def test_frames_no_encoder_writes_are_answered_like_libzstd(zstd):
    """tests/craft.py: repeat offset 1 minus one = 0 (zstd.c:46941), blocks above the frame's block maximum in libzstd's one-pass and
    streaming decoders (zstd.c:44239-44246, :47714) -- accepted exactly when libzstd accepts, with libzstd's bytes; alone (K1 / K2 / K3 or
    the generic kernel) and all together in one batch."""
    from tests import craft, reflib
    cases = craft.edge_frames() + craft.skippable_frames() + craft.encoding_variants()      # (skippable frames as items: passed over, an empty segment; c-ext/decompressor.c:986-988 reads their size 0)
    ref = reflib.RefZstd() if reflib.have_ref() else None
    d = zstd.ZstdDecompressor()
    good = []
    for name, f, n, ok in cases:
        want = None
        if ref is not None:
            try: want = ref.decompress(f, n)
            except RuntimeError: want = None
            assert (want is not None) == ok, name
        sizes = struct.pack("=Q", n)
        if ok and n == 0:                       # nothing produced: next to a frame that produces something (a result without bytes is refused, c-ext/bufferutil.c:408-412)
            res = d.multi_decompress_to_buffer([cases[1][1], f], decompressed_sizes=struct.pack("=2Q", cases[1][2], 0))
            assert len(res[0].tobytes()) == cases[1][2] and res[1].tobytes() == b"", name
            good.append((f, 0, b""))
        elif ok:
            got = d.multi_decompress_to_buffer([f], decompressed_sizes=sizes)[0].tobytes()
            assert len(got) == n and (want is None or got == want), name
            good.append((f, n, got))
        elif n == 0:                            # the size comes from the header, and the header is what is wrong (c-ext/decompressor.c:990-993)
            with pytest.raises((ValueError, zstd.ZstdError), match="could not determine decompressed size of item 0|error decompressing item 0"):
                d.multi_decompress_to_buffer([f], decompressed_sizes=sizes)
            with pytest.raises(zstd.ZstdError, match="error decompressing item 1"):
                d.multi_decompress_to_buffer([cases[1][1], f], decompressed_sizes=struct.pack("=2Q", cases[1][2], 7))
        else:
            with pytest.raises(zstd.ZstdError, match="error decompressing item 0"):
                d.multi_decompress_to_buffer([f], decompressed_sizes=sizes)
    res = d.multi_decompress_to_buffer([g[0] for g in good] * 3, decompressed_sizes=struct.pack("=%dQ" % (3 * len(good)), *([g[1] for g in good] * 3)))
    assert [res[i].tobytes() for i in range(len(res))] == [g[2] for g in good] * 3


def test_frames_of_several_blocks_take_the_phase_split_kernels(zstd, corpus):
    """Batches whose items exceed 128 KiB run the pipeline's several-block mode (zhip_format.hpp ZpFrameRec; the host API's size hint turns it
    on): libzstd frames of 1-8 blocks at several levels with and without checksum next to single-block ones; hand-made frames of many tiny
    blocks that exhaust the chunk's block slots (those are the generic kernel's); a frame damaged in a middle block is the item reported.
    Reference behaviour: ZSTD_decompressFrame's block loop, zstd/zstd.c:44207-44262."""
    from tests import craft, reflib
    if not reflib.have_ref():
        pytest.fail("no libzstd 1.5.7 to check against: " + "needs reference libzstd for frames of other levels")
    ref = reflib.RefZstd()
    rng = np.random.default_rng(77)
    raws = []
    for i in range(40):
        n = int(rng.choice([rng.integers(1, 5000), rng.integers(100000, 131073), rng.integers(131073, 1048577)]))
        raws.append(b"".join(corpus.frame_bytes(400 + 8 * i + k) for k in range(n // 131072 + 1))[:n])
    raws += [rng.bytes(400000), b"\x05" * 700000, (corpus.frame_bytes(9)[:700] + rng.bytes(90)) * 900]
    frames = [ref.compress(r, level=int(rng.choice([1, 3, 3, 3, 5, 9])), flags=7 if i % 3 == 0 else 5) for i, r in enumerate(raws)]
    tiny = [bytes([65 + (k % 26)]) * 9 for k in range(120)]                                   # 120 raw blocks of nine bytes
    many = craft._hdr(9 * 120 + 12, 17) + b"".join(craft.raw_block(t) for t in tiny) + craft.sequences_block(b"xy", [(1, 4, 2), (0, 3, 2), (1, 3, 1)])
    want_many = ref.decompress(many, 9 * 120 + 12)
    assert want_many[:1080] == b"".join(tiny)
    frames += [many] * 60; raws += [want_many] * 60
    d = zstd.ZstdDecompressor()
    sizes = struct.pack("=%dQ" % len(raws), *map(len, raws))
    res = d.multi_decompress_to_buffer(frames, decompressed_sizes=sizes)
    for i, r in enumerate(raws):
        assert res[i].tobytes() == r, "frame %d (%d bytes)" % (i, len(r))
    k = max(range(0, 40, 3), key=lambda i: len(frames[i]))                                    # the largest frame with a checksum: damage a byte two thirds in
    bad = bytearray(frames[k]); bad[2 * len(bad) // 3] ^= 0x55
    try: ref.decompress(bytes(bad), len(raws[k])); pytest.skip("the damage went unnoticed by libzstd")
    except RuntimeError: pass
    broken = list(frames); broken[k] = bytes(bad)
    with pytest.raises(zstd.ZstdError, match="error decompressing item %d" % k):
        d.multi_decompress_to_buffer(broken, decompressed_sizes=sizes)


def test_content_checksum_is_verified(zstd):
    from tests import reflib
    if not reflib.have_ref():
        pytest.fail("no libzstd 1.5.7 to check against: " + "needs reference libzstd for checksum frames")
    ref = reflib.RefZstd()
    from tests.corpus import Corpus
    c = Corpus()
    raws = [b"a" * 1000, c.frame_bytes(7), c.frame_bytes(8)[:5000], b"".join(c.frame_bytes(i) for i in range(2))]
    frames = [ref.compress(r, flags=reflib.DEFAULT_FLAGS | reflib.F_CHECKSUM) for r in raws]
    res = zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
    assert [res[i].tobytes() for i in range(len(raws))] == raws
    for k in range(len(frames)):
        bad = list(frames)
        bad[k] = bad[k][:-2] + bytes([bad[k][-2] ^ 0x20]) + bad[k][-1:]
        with pytest.raises(zstd.ZstdError, match="error decompressing item %d: Restored data doesn't match checksum" % k):
            zstd.ZstdDecompressor().multi_decompress_to_buffer(bad)


def test_decompress_content_dict_chain(zstd, oracle):
    """ZstdDecompressor.decompress_content_dict_chain (c-ext/decompressor.c:620-890): every frame after the first is
    decoded with the previous fulltext as a raw-content dictionary; the reference's own test for it runs in tests/run_reference_hotpath_tests.sh"""
    from tests.corpus import Corpus
    c = Corpus()
    original = [b"foo" * 64, b"foobar" * 64, b"baz" * 64, b"foobaz" * 64, b"foobarbaz" * 64, c.frame_bytes(7)[:9000], c.frame_bytes(7)[400:9400]]
    chunks = [zstd.ZstdCompressor().compress(original[0])]
    for i, chunk in enumerate(original[1:]):
        chunks.append(zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(original[i])).compress(chunk))
    d = zstd.ZstdDecompressor()
    for i in range(1, len(original) + 1):
        assert d.decompress_content_dict_chain(chunks[0:i]) == original[i - 1]
    assert len(chunks[6]) < len(zstd.ZstdCompressor().compress(original[6])) // 4           # the chain's dictionary really is used
    with pytest.raises(zstd.ZstdError, match="chunk 1 did not decompress full frame"):
        d.decompress_content_dict_chain([chunks[0], chunks[1][0:12] + chunks[1][15:]])
    with pytest.raises(ValueError, match="chunk 1 missing content size in frame"):
        d.decompress_content_dict_chain([chunks[0], zstd.ZstdCompressor(write_content_size=False).compress(b"foo" * 64)])


def test_dictionary_frames_through_the_pipeline(zstd, corpus):
    """dictionary frames take the phase-split pipeline since r02v (ready-made dictionary tables for treeless literals / repeat-mode sequence
    tables, repeat offsets from the dictionary, match sources in its content incl. items that straddle the frame's first byte): 1 024
    documents per dictionary against libzstd's frames, a frame made without the dictionary in the same batch, a wrong / missing dictionary"""
    import numpy as np
    from tests import reflib
    if not reflib.have_ref():
        pytest.fail("no libzstd 1.5.7 to check against: " + "needs oracle/_ref (libzstd 1.5.7) for the frames")
    ref = reflib.RefZstd()
    rng = np.random.default_rng(7)
    docs = [corpus.frame_bytes(700 + i)[j * 4096:(j + 1) * 4096] for i in range(32) for j in range(32)]
    trained = ref.train_dictionary(16384, [f[:3000] for f in corpus.frame_list(900, 400)])
    rawd = corpus.frame_bytes(600)[:6000]
    for dd in (trained, rawd):
        content = dd[-4000:]
        b30 = rng.bytes(30)
        straddlers = [b30 + content[-20:] + b30[:15] + rng.bytes(10) + content[-40:] + b30[:25],
                      content[-300:] + content[-300:] + rng.bytes(5) + content[-64:] + content[-300:-250]]
        raws = docs + [b"a", corpus.frame_bytes(5)[:40000], (dd[-3000:] + docs[3])[:6000]] + straddlers
        for level in (3, 1):
            frames = [ref.compress(r, level=level, dict_data=dd) for r in raws] + [ref.compress(docs[0], level=level)]
            got = zstd.ZstdDecompressor(dict_data=zstd.ZstdCompressionDict(dd)).multi_decompress_to_buffer(frames)
            assert len(got) == len(frames)
            for i, r in enumerate(raws + [docs[0]]):
                assert got[i].tobytes() == r, (level, i)
    other = ref.train_dictionary(8192, [f[:2000] for f in corpus.frame_list(300, 300)])
    frame = ref.compress(docs[1], level=3, dict_data=trained)
    for d in (zstd.ZstdDecompressor(dict_data=zstd.ZstdCompressionDict(other)), zstd.ZstdDecompressor()):
        with pytest.raises(zstd.ZstdError, match="Dictionary mismatch"):
            d.multi_decompress_to_buffer([frame], decompressed_sizes=np.array([4096], dtype=np.uint64).tobytes())


def test_understated_size_hint_runs_out_of_room_gracefully(zstd, corpus):
    """Round 5: the decode pipeline's compact arena is sized from the caller's size HINT (zhip_ctx_set_size_hint: 4 x the hint + 1 KiB of literal + sequence
    room per frame, three chunk slots for small frames). The hint is advisory: 2 048 frames of 128 KiB behind a hint of 4 KiB overrun that budget many
    times over -- the frames that find no room must come back from the generic kernel, every byte right, none of the others disturbed."""
    import importlib
    import torch
    dev_mod = importlib.import_module("zstandard_amd.device")
    enc = _enc()
    dev = torch.device("cuda", 0)
    F, item = 2048, 131072
    raws = [corpus.frame_bytes(i % 600) for i in range(F)]
    frames = [enc.compress(r) for r in raws[:600]]
    frames = [frames[i % 600] for i in range(F)]
    sizes = np.array([len(f) for f in frames], dtype=np.int64)
    offs = np.zeros(F, dtype=np.int64); offs[1:] = np.cumsum(sizes)[:-1]
    def segs(o, n):
        a = np.zeros((F, 2), dtype=np.int64); a[:, 0] = o; a[:, 1] = n
        return torch.from_numpy(a).to(dev)
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(F * item, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev); status = torch.zeros(F, dtype=torch.int32, device=dev)
    want = torch.from_numpy(np.frombuffer(b"".join(raws[:600]), dtype=np.uint8).copy()).view(600, item).to(dev)
    for hint in (4096, 0):                                      # understated, then none (the default budget: everybody has room)
        ctx = dev_mod.DeviceBatchContext()
        try:
            ctx.set_size_hint(hint)
            dst.zero_()
            ctx.decompress(src, segs(offs, sizes), dst, segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64)), out_sizes, status)
            torch.cuda.synchronize()
            assert int(status.abs().max().item()) == 0 and bool((out_sizes == item).all().item()), hint
            got = dst.view(F, item)
            for a in range(0, F, 600):
                b = min(F, a + 600)
                assert torch.equal(got[a:b], want[: b - a]), (hint, a)
            if hint:
                assert ctx.kernel_time(0) is not None           # (the generic kernel's timer exists; its share is what the hint cost)
        finally:
            ctx.close()


def test_k0_on_small_and_unusual_batches(zstd, corpus):
    """K0 -- the lane-per-frame pass that walks the Huffman weights' description and the three sequence distributions for K1 (round 6) -- runs from 6 144 frames per chunk on, which
    this suite's batches never reach. ZHIP_K0_MIN=0 (read when a context is created) turns it on for every batch: a mixed batch -- frames of several levels and shapes, the hand-made
    frames no encoder writes, skippable frames, raw / RLE blocks, damaged and truncated copies, a frame of several blocks, an empty frame -- through a context with K0 and one without
    must give the same status and the same bytes frame by frame, and libzstd's bytes wherever libzstd decodes."""
    import os
    import torch
    from zstandard_amd.device import DeviceBatchContext
    from tests import craft, reflib
    ref = reflib.checker()
    rng = np.random.default_rng(91)
    raws = []
    for i in range(60):
        base = corpus.frame_bytes(1000 + i)
        n = int(rng.integers(200, 131073)); k = i % 6
        r = (base[:n] if k == 0 else bytes(rng.integers(0, 12, n, dtype=np.uint8)) if k == 1 else (base[:300] + bytes(rng.integers(97, 105, 80, dtype=np.uint8))) * (n // 380 + 1) if k == 2
             else bytes((np.frombuffer(base[:n], dtype=np.uint8) & 0x3F).tobytes()) if k == 3 else rng.bytes(n // 8) + base[:n] if k == 4 else base[:n // 2] + bytes(n // 2))
        raws.append(r[:n])
    levels = [3, 1, 3, 5, 3, 9, 3, -1, 19]
    frames = [ref.compress(r, level=levels[i % len(levels)]) for i, r in enumerate(raws)]
    sizes = [len(r) for r in raws]
    cases = craft.edge_frames() + craft.skippable_frames() + craft.encoding_variants()
    frames += [c[1] for c in cases]; sizes += [c[2] for c in cases]
    frames += [ref.compress(b""), ref.compress(corpus.frame_bytes(5) + corpus.frame_bytes(6)[:70000])]; sizes += [0, 131072 + 70000]
    for k in range(80):
        f = bytearray(frames[k % 60])
        if k % 8 == 7:
            f = f[:max(6, len(f) - 1 - k % 5)]
        else:
            for _ in range(1 + k % 2):
                f[int(rng.integers(5, min(len(f), 220)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bytes(f)); sizes.append(sizes[k % 60])
    n = len(frames)
    dev = torch.device("cuda", 0)
    csz = np.array([len(f) for f in frames], dtype=np.int64)
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    ss = np.zeros((n, 2), dtype=np.int64); ss[1:, 0] = np.cumsum(csz)[:-1]; ss[:, 1] = csz
    cap = np.array([s + 64 for s in sizes], dtype=np.int64)
    ds = np.zeros((n, 2), dtype=np.int64); ds[1:, 0] = np.cumsum(cap)[:-1]; ds[:, 1] = np.array(sizes, dtype=np.int64)
    res = {}
    for k0 in ("off", "on"):
        if k0 == "on":
            os.environ["ZHIP_K0_MIN"] = "0"
        try:
            ctx = DeviceBatchContext()
        finally:
            os.environ.pop("ZHIP_K0_MIN", None)
        dst = torch.full((int(cap.sum()),), 0xA5, dtype=torch.uint8, device=dev)
        out_sizes = torch.zeros(n, dtype=torch.int64, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
        ctx.decompress(src, torch.from_numpy(ss).to(dev), dst, torch.from_numpy(ds).to(dev), out_sizes, status)
        torch.cuda.synchronize()
        res[k0] = (status.cpu().numpy().copy(), out_sizes.cpu().numpy().copy(), dst.cpu().numpy().copy())
        ctx.close()
    st0, os0, d0 = res["off"]; st1, os1, d1 = res["on"]
    assert (st0 == st1).all(), [(i, int(a), int(b)) for i, (a, b) in enumerate(zip(st0, st1)) if a != b][:8]
    assert (os0 == os1).all()
    good = 0
    for i in range(n):
        if st1[i] == 0:
            o = int(ds[i, 0]); got = d1[o:o + int(os1[i])].tobytes()
            assert got == d0[o:o + int(os0[i])].tobytes(), i
            try:
                want = ref.decompress(frames[i], sizes[i])
            except RuntimeError:
                want = None
            assert want is not None and got == want, i
            good += 1
    assert good >= 70 and int((st1 != 0).sum()) >= 30, (good, int((st1 != 0).sum()))


def test_frames_cut_by_the_block_splitter_take_the_several_block_mode(zstd, corpus):
    """libzstd's block splitter (levels 16 and up) cuts even a 128 KiB source into many blocks: such frames are frames of several blocks although no larger than one. The host-buffer
    call counts every frame's blocks from its block headers and gives a chunk with enough of them to the pipeline's several-block mode (round 6: one wave each in the generic kernel,
    they decoded at 6 GB/s). Level-19 and level-17 frames, alone, mixed with level-3 frames, and a few among many level-3 frames (those few stay the generic kernel's): every byte back."""
    from tests import reflib
    ref = reflib.checker()
    raws = [corpus.frame_bytes(1300 + i)[: 131072 - 997 * (i % 5)] for i in range(96)]
    hi = [ref.compress(r, level=19 if i % 2 else 17) for i, r in enumerate(raws)]
    lo = [ref.compress(r, level=3) for r in raws]
    blocks = []
    for f in hi[:8]:
        fhd = f[4]; single = (fhd >> 5) & 1; pos = 5 + (0 if single else 1) + [0, 1, 2, 4][fhd & 3] + ([1 if single else 0, 2, 4, 8][fhd >> 6]); nb = 0
        while True:
            bh = int.from_bytes(f[pos:pos + 3], "little"); nb += 1; pos += 3 + (1 if (bh >> 1) & 3 == 1 else bh >> 3)
            if bh & 1: break
        blocks.append(nb)
    assert max(blocks) > 1, blocks                       # (the splitter did cut them)
    d = zstd.ZstdDecompressor()
    for batch, want in ((hi, raws), (hi[:40] + lo[:40], raws[:40] + raws[:40]), (lo + hi[:3], raws + raws[:3]), ([hi[5]], [raws[5]])):
        res = d.multi_decompress_to_buffer(batch)
        assert len(res) == len(batch) and all(res[i].tobytes() == want[i] for i in range(len(batch)))
    assert d.decompress(hi[7]) == raws[7]
