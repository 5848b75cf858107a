"""GPU parity tests for the boundary fields round 1 left out of the C ABI: dictionary content type, frame format,
explicit compression parameters -- each against the reference build (oracle/_ref) -- plus a BASELINE-shaped batch of >= 8192 frames
per direction compared frame by frame, and hostile frame headers (the sizes a frame claims are untrusted input)."""
import hashlib
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zstd():
    import zstandard_amd
    assert zstandard_amd._lib.lib().zhip_device_count() >= 1, "no GPU visible"
    return zstandard_amd


def test_a_whole_block_as_one_match_into_the_dictionary(zstd, ref):
    """A 128 KiB source that IS its (raw-content) dictionary compresses to one sequence of match length 131 072 -- beyond 17 bits: the
    packed sequences K2 hands to K3 give the match length 18 (round 3: the 17-bit form dropped the top bit and the frame was refused).
    Both directions, with the neighbours one and two bytes short and a frame whose match starts after a literal."""
    rng = np.random.default_rng(5)
    data = rng.bytes(131072)
    blob = b"zz" + data
    raws = [data, data[:131071], data[:131070], data[:70000], data[:3] + b"Q" + data[:131068]] * 3
    zd = zstd.ZstdCompressionDict(blob, dict_type=zstd.DICT_TYPE_RAWCONTENT)
    res = zstd.ZstdCompressor(level=3, dict_data=zd).multi_compress_to_buffer(raws)
    frames = [ref.compress_advanced(r, level=3, dict_data=blob, dict_type=zstd.DICT_TYPE_RAWCONTENT) for r in raws]
    assert [res[i].tobytes() for i in range(len(raws))] == frames and len(frames[0]) < 40
    back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(frames)
    assert [back[i].tobytes() for i in range(len(raws))] == raws
    assert zstd.ZstdDecompressor(dict_data=zd).decompress(frames[0]) == data


def test_dictionary_content_type(zstd, ref, corpus):
    """c-ext/compressiondict.c:170-191 -> compressor.c:37-52 / compressiondict.c:148-162: DICT_TYPE_RAWCONTENT treats a blob that starts
    with the dictionary magic as plain content (different frames from AUTO!), DICT_TYPE_FULLDICT demands the magic"""
    from tests.test_oracle_vs_golden import _dict_vectors
    dicts, _ = _dict_vectors()
    trained = dicts["trained"]                                            # starts with 37 a4 30 ec
    assert trained[:4] == bytes.fromhex("37a430ec")
    fake = bytes.fromhex("37a430ec") + b"not an entropy section at all " * 40     # magic but no valid tables
    raws = [corpus.frame_bytes(40 + i)[300:300 + 1500 + 700 * i] for i in range(12)] + [b"tiny"]
    for blob, dtype, label in ((trained, zstd.DICT_TYPE_RAWCONTENT, "trained as raw"), (trained, zstd.DICT_TYPE_FULLDICT, "trained as full"),
                               (trained, zstd.DICT_TYPE_AUTO, "trained auto"), (fake, zstd.DICT_TYPE_RAWCONTENT, "magic-prefixed content as raw"),
                               (b"short", zstd.DICT_TYPE_AUTO, "5-byte dictionary (not loaded at all)"),
                               (corpus.frame_bytes(60)[:5000], zstd.DICT_TYPE_RAWCONTENT, "content as raw")):
        zd = zstd.ZstdCompressionDict(blob, dict_type=dtype)
        res = zstd.ZstdCompressor(level=3, dict_data=zd).multi_compress_to_buffer(raws)
        for i, r in enumerate(raws):
            assert res[i].tobytes() == ref.compress_advanced(r, level=3, dict_data=blob, dict_type=dtype), (label, i)
        back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(res)
        assert [back[i].tobytes() for i in range(len(raws))] == raws, label
        assert ref.decompress_advanced(res[3].tobytes(), len(raws[3]), dict_data=blob, dict_type=dtype) == raws[3]
    # raw and auto readings of the same blob are different frames (the bug round 1 had: dict_type never reached the kernels)
    a = zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(trained)).compress(raws[2])
    b = zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(trained, dict_type=zstd.DICT_TYPE_RAWCONTENT)).compress(raws[2])
    assert a != b and zstd.get_frame_parameters(a).dict_id != 0 and zstd.get_frame_parameters(b).dict_id == 0
    # a full dictionary is demanded but the blob is not one, or the blob is damaged: libzstd's errors, both directions. On the compression
    # side that is "Allocation error" whatever the cause: the CDict is created lazily (ZSTD_initLocalDict, zstd.c:24206) and a NULL result
    # is all the caller learns (zstd.c:24232); the reference's message is "cannot compress: ..." (c-ext/compressor.c:560)
    for blob in (b"plain content without magic " * 10, b"abc"):
        with pytest.raises(RuntimeError, match="Allocation error"):
            ref.compress_advanced(raws[0], dict_data=blob, dict_type=2)
        with pytest.raises(zstd.ZstdError, match="cannot compress: Allocation error : not enough memory"):
            zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(blob, dict_type=zstd.DICT_TYPE_FULLDICT)).compress(raws[0])
        with pytest.raises(zstd.ZstdError, match="error compressing item 0: Allocation error"):
            zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(blob, dict_type=zstd.DICT_TYPE_FULLDICT)).multi_compress_to_buffer(raws)
        with pytest.raises(zstd.ZstdError, match="could not create decompression dict"):
            zstd.ZstdDecompressor(dict_data=zstd.ZstdCompressionDict(blob, dict_type=zstd.DICT_TYPE_FULLDICT)).decompress(a)
    with pytest.raises(RuntimeError, match="Allocation error"):                           # magic, then garbage, read as a full dictionary
        ref.compress_advanced(raws[0], dict_data=fake, dict_type=0)
    with pytest.raises(zstd.ZstdError, match="cannot compress: Allocation error"):
        zstd.ZstdCompressor(dict_data=zstd.ZstdCompressionDict(fake)).compress(raws[0])


def test_compression_parameters_and_format(zstd, ref, corpus):
    """ZstdCompressor(compression_params=ZstdCompressionParameters(...)) -- c-ext/compressor.c:203 -> set_parameters: explicit fields,
    from_level() rows, the flags carried by the object, the magicless format in both directions"""
    P = zstd.ZstdCompressionParameters
    raws = [corpus.frame_bytes(300 + i)[: 2000 + 9000 * i] for i in range(14)] + [corpus.frame_bytes(320), b"foobar"]
    cases = [dict(hash_log=12, chain_log=10), dict(min_match=6), dict(strategy=zstd.STRATEGY_FAST, target_length=3), dict(strategy=zstd.STRATEGY_DFAST, min_match=7),
             dict(compression_level=1), dict(compression_level=-5), dict(window_log=17, hash_log=17, chain_log=16, search_log=1, min_match=5, target_length=0, strategy=2)]
    for kw in cases:
        params = P(write_checksum=1, write_dict_id=1, **kw)
        res = zstd.ZstdCompressor(compression_params=params).multi_compress_to_buffer(raws)
        rkw = {k: v for k, v in kw.items() if k != "compression_level"}
        for i, r in enumerate(raws):
            assert res[i].tobytes() == ref.compress_advanced(r, level=kw.get("compression_level", 3), flags=7, **rkw), (kw, i)
    # a window of exactly one block over several full blocks: every later block starts with an EMPTY prefix (zstd.c:31091 `ip += (dictAndPrefixLength == 0)`,
    # found by tests/stress_emu_params.py seed 740)
    text = b"".join(corpus.frame_list(40, 4))
    blk = corpus.frame_bytes(3)[:700] + corpus.frame_bytes(4)[:2076]
    several = [text[:300000], (blk * 120)[:262144], text[:262144], (blk * 200)[:400000]]
    for strat in (zstd.STRATEGY_FAST, zstd.STRATEGY_DFAST):
        res = zstd.ZstdCompressor(compression_params=P(window_log=17, strategy=strat)).multi_compress_to_buffer(several)
        for i, r in enumerate(several):
            assert res[i].tobytes() == ref.compress_advanced(r, level=3, flags=1, window_log=17, strategy=strat), (strat, i)
    # from_level: the row of an unknown-size source made explicit -- not the same frames as level=3 (which picks the row per source size)
    fl = zstd.ZstdCompressor(compression_params=P.from_level(3)).multi_compress_to_buffer(raws)
    kw = dict(window_log=21, chain_log=16, hash_log=17, search_log=1, min_match=5, target_length=0, strategy=2)
    for i, r in enumerate(raws):
        assert fl[i].tobytes() == ref.compress_advanced(r, level=3, flags=1, **kw), i             # the object's write_dict_id defaults to 0
    assert zstd.ZstdCompressor(compression_params=P.from_level(3)).compress(b"") == bytes.fromhex("28b52ffd2000010000")
    # strategies the kernels do not implement: loud
    with pytest.raises(zstd.ZstdError):
        zstd.ZstdCompressor(compression_params=P.from_level(9)).compress(raws[3])
    # magicless frames (reference tests test_no_magic / test_headerless)
    m = zstd.ZstdCompressor(compression_params=P.from_level(1, format=zstd.FORMAT_ZSTD1_MAGICLESS))
    z = zstd.ZstdCompressor(compression_params=P.from_level(1, format=zstd.FORMAT_ZSTD1))
    assert z.compress(b"foobar")[:4] == b"\x28\xb5\x2f\xfd" and z.compress(b"foobar")[4:] == m.compress(b"foobar")
    res = m.multi_compress_to_buffer(raws)
    kw1 = dict(zip(("window_log", "chain_log", "hash_log", "search_log", "min_match", "target_length", "strategy"), (19, 13, 14, 1, 7, 0, 1)))
    for i, r in enumerate(raws):
        assert res[i].tobytes() == ref.compress_advanced(r, level=3, flags=1, format=1, **kw1), i
    d = zstd.ZstdDecompressor(format=zstd.FORMAT_ZSTD1_MAGICLESS)
    back = d.multi_decompress_to_buffer(res)
    assert [back[i].tobytes() for i in range(len(raws))] == raws and d.decompress(res[5].tobytes()) == raws[5]
    with pytest.raises(zstd.ZstdError, match="error determining content size from frame header"):
        zstd.ZstdDecompressor().decompress(res[5].tobytes())
    assert zstd.get_frame_parameters(res[14].tobytes(), format=zstd.FORMAT_ZSTD1_MAGICLESS).content_size == len(raws[14])


def test_large_batch_both_directions_frame_by_frame(zstd, ref):
    """>= 8192 frames per direction: BASELINE-shaped 128 KiB inputs plus ragged ones, every compressed frame
    compared with the reference build's (sha256), every decoded frame with its source"""
    import torch
    from tests.corpus import Corpus
    dev = torch.device("cuda", 0)
    n_full, n_ragged = 6144, 2560
    raw = Corpus(device=dev).frames(70000, n_full, chunk=256).cpu().numpy()
    rng = np.random.default_rng(12)
    items = [raw[i].tobytes() for i in range(n_full)]
    items += [raw[int(rng.integers(0, n_full))][: int(rng.integers(1, 131073))].tobytes() for _ in range(n_ragged)]
    want = [ref.compress(r) for r in items]
    got = zstd.ZstdCompressor(level=3).multi_compress_to_buffer(items)
    assert len(got) == len(items)
    for i in range(len(items)):
        f = got[i].tobytes()
        assert len(f) == len(want[i]) and hashlib.sha256(f).digest() == hashlib.sha256(want[i]).digest(), "compressed frame %d differs from libzstd" % i
    back = zstd.ZstdDecompressor().multi_decompress_to_buffer(want)
    assert len(back) == len(items) and back.size() == sum(map(len, items))
    for i in range(len(items)):
        assert back[i].tobytes() == items[i], "decoded frame %d" % i


def test_hostile_content_sizes(zstd):
    """the decompressed size a frame header claims is untrusted: claims that cannot be allocated are MemoryError /
    ZstdError, never a wrapped allocation that the kernels then write past"""
    def frame(fcs, payload_blocks):
        # single-segment off, 8-byte frame content size, then raw/RLE blocks
        return bytes.fromhex("28b52ffd") + bytes([0xC0, 0x38]) + struct.pack("<Q", fcs) + payload_blocks
    rle = lambda n, last: struct.pack("<I", (n << 3) | 2 | last)[:3] + b"z"
    huge = frame((1 << 64) - 8, rle(100000, 0) * 8 + rle(100000, 1))
    half = frame(1 << 63, rle(5, 1))
    good = zstd.ZstdCompressor().compress(b"ok" * 500)
    d = zstd.ZstdDecompressor(max_window_size=1 << 31)
    for batch in ([huge], [half, half], [good, huge], [half, good, half]):
        with pytest.raises((MemoryError, zstd.ZstdError)):
            d.multi_decompress_to_buffer(batch)
    with pytest.raises((MemoryError, zstd.ZstdError)):
        d.decompress(huge)
    # a large-but-allocatable lie is caught by the kernels' own bounds: the frame produces 800 005 bytes, claims 64 MiB
    lie = frame(64 << 20, rle(100000, 0) * 8 + rle(5, 1))
    with pytest.raises(zstd.ZstdError, match="error decompressing item 0"):
        d.multi_decompress_to_buffer([lie])
    assert d.multi_decompress_to_buffer([good])[0].tobytes() == b"ok" * 500             # the context is still healthy


def test_dictionary_batch_config4_shape(zstd):
    """BASELINE.json configs[3] at a sixteenth of its document count and at its full dictionary: 16 384 x 4 KiB JSON-like documents with
    the 112 640-byte dictionary train_dictionary(112640, 10 000 samples) makes (tests/golden/dict_json4k.bin, SURVEY.md 8(d)4: a CDict of
    W17 / C15 / H16 = 384 KiB of tagged tables and 110 KiB of content behind every frame) -- every compressed frame equals the reference
    build's (native ZSTD_CCtx_refCDict workers, the reference's own call), every frame decodes back with the dictionary; the pinned probe
    vectors of the fixture; and sources that straddle into / reach deep inside the 110 KiB of dictionary content, both directions"""
    import json
    import os
    import bench
    from tests import reflib
    from tests.corpus import Corpus
    if not reflib.have_ref():
        pytest.fail("no libzstd 1.5.7 to check against: " + "needs the reference build")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = open(os.path.join(root, "tests", "golden", "dict_json4k.bin"), "rb").read()
    meta = json.load(open(os.path.join(root, "tests", "golden", "dict_json4k.json")))
    assert hashlib.sha256(blob).hexdigest() == meta["dict_sha256"]
    n = 16384
    docs = Corpus(frame_size=4096).json_docs(0, n).numpy()
    assert [hashlib.sha256(docs[i].tobytes()).hexdigest() for i in range(8)] == meta["probe_docs_sha256"]
    want, sizes = bench.compress_on_host(docs, 4096, blob)
    zd = zstd.ZstdCompressionDict(blob)
    items = [docs[i].tobytes() for i in range(n)]
    got = zstd.ZstdCompressor(level=3, dict_data=zd).multi_compress_to_buffer(items)
    assert len(got) == n
    for i in range(n):
        f = got[i].tobytes()
        assert len(f) == len(want[i]) and f == want[i], "document %d: frame differs from libzstd's" % i
    assert [hashlib.sha256(got[i].tobytes()).hexdigest() for i in range(8)] == meta["probe_frames_sha256"]
    back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(got)
    assert len(back) == n and back.size() == n * 4096
    for i in range(n):
        assert back[i].tobytes() == items[i], "document %d does not round-trip" % i
    # straddlers: sources that continue the dictionary's last bytes, quote pieces from its start / middle / end (offsets up to ~110 KiB
    # below the frame's first byte), tiny and empty sources, and sizes on both sides of the attach cutoff (16 KiB for double-fast)
    assert len(blob) == 112640 and meta["dict_size"] == 112640
    ref = reflib.RefZstd()
    content = blob[-100000:]
    extra = [content[-3000:] + items[0][:1000], content[5000:6500] + items[1][:2000] + content[60000:61000], content[-100:], b"", b"x",
             content[:4096], content[40000:44096], items[2][:100] + content[-50000:-46000] + items[3][:100],
             content[-16384:], content[-16385:] + b"!", content[1000:1000 + 16000] + items[4][:384], (content[-700:] * 30)[:16384], items[5] * 4, items[6] * 5]
    got2 = zstd.ZstdCompressor(level=3, dict_data=zd).multi_compress_to_buffer(extra)
    for i, r in enumerate(extra):
        assert got2[i].tobytes() == ref.compress(r, level=3, dict_data=blob), "straddler %d: frame differs from libzstd's" % i
    back2 = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(got2)
    assert [back2[i].tobytes() for i in range(len(extra))] == extra
    # frames made by libzstd at other levels against the same dictionary (long offsets into the content) decode identically
    deep = [ref.compress(r, level=lv, dict_data=blob) for lv in (1, 3, 9, 19) for r in extra[:8]]
    back3 = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(deep)
    assert [back3[i].tobytes() for i in range(len(deep))] == [r for lv in (1, 3, 9, 19) for r in extra[:8]]


def test_fast_strategy_with_dictionary(zstd, ref, corpus):
    """ZstdCompressor(level=1, dict_data=...) -- what six of the reference's own hot-path tests do (test_compress_dict_multiple,
    test_dict_precompute, test_no_dict_id, test_dictionary*, test_dict): the fast strategy's dictionary search
    (ZSTD_compressBlock_fast_dictMatchState_generic, zstd.c:32197), frames equal to the reference build's, round trip with the dictionary"""
    samples = []
    for i in range(128):
        samples += [b"foo" * 64, b"bar" * 64, b"foobar" * 64]
    trained = ref.train_dictionary(8192, samples)
    raws = [corpus.frame_bytes(700 + i)[: 300 + 257 * i] for i in range(30)] + [b"foo bar foobar foo bar foobar", b"foobar" * 1000, b"x"]
    for blob in (trained, corpus.frame_bytes(600)[:6000]):
        zd = zstd.ZstdCompressionDict(blob)
        for level in (1, 2, -5):
            got = zstd.ZstdCompressor(level=level, dict_data=zd).multi_compress_to_buffer(raws)
            for i, r in enumerate(raws):
                assert got[i].tobytes() == ref.compress(r, level=level, dict_data=blob), (level, i)
            back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(got)
            assert [back[i].tobytes() for i in range(len(raws))] == raws
        assert zstd.ZstdCompressor(level=1, dict_data=zd).compress(raws[3]) == ref.compress(raws[3], level=1, dict_data=blob)
    # above libzstd's attach cutoffs (8 KiB fast / 16 KiB double-fast, zstd.c:25250) it copies the dictionary's tables and searches the
    # content as an external segment (ZSTD_resetCCtx_byCopyingCDict + the _extDict searches): same frames here, up to one block
    big = [b"foobar" * 16384, corpus.frame_bytes(5)[:9000], corpus.frame_bytes(6)[:16385], corpus.frame_bytes(7)[:70000], corpus.frame_bytes(8)]
    for blob in (trained, corpus.frame_bytes(600)[:6000]):
        zd = zstd.ZstdCompressionDict(blob)
        for level in (1, 3):
            got = zstd.ZstdCompressor(level=level, dict_data=zd).multi_compress_to_buffer(big + raws[:4])
            for i, r in enumerate(big + raws[:4]):
                assert got[i].tobytes() == ref.compress(r, level=level, dict_data=blob), ("copy mode", level, i)
            back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(got)
            assert [back[i].tobytes() for i in range(len(big) + 4)] == big + raws[:4]
    # several blocks against a dictionary (the reference's generate_samples() reaches 196 608 bytes)
    several = [corpus.frame_bytes(5) + b"tail", b"baz" * 65536, b"".join(corpus.frame_bytes(20 + i) for i in range(3)), (b"".join(corpus.frame_bytes(30 + i) for i in range(4)))[:1 << 19]]
    for level in (1, 3):
        zd = zstd.ZstdCompressionDict(trained)
        got = zstd.ZstdCompressor(level=level, dict_data=zd).multi_compress_to_buffer(several)
        for i, r in enumerate(several):
            assert got[i].tobytes() == ref.compress(r, level=level, dict_data=trained), ("multi-block", level, i)
        back = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(got)
        assert [back[i].tobytes() for i in range(len(several))] == several
    with pytest.raises(zstd.ZstdError):                                           # a source larger than its window (level 1: 512 KiB): libzstd drops the
        zstd.ZstdCompressor(level=1, dict_data=zstd.ZstdCompressionDict(trained)).compress(b"q" * ((1 << 19) + 1))   # dictionary part-way; refused, loudly


def test_precomputed_dictionary(zstd, ref, corpus):
    """ZstdCompressionDict.precompute_compress (c-ext/compressiondict.c:228-286 -> ZSTD_createCDict_advanced; c-ext/compressor.c:29-31,1147
    ZSTD_CCtx_refCDict): frames made with a precomputed dictionary follow the DICTIONARY's level / parameters, not the compressor's --
    bit-identical to libzstd driven the same way, in attach mode, table-copy mode and over several blocks; argument errors and a
    dictionary libzstd refuses to digest are reported as the reference reports them; .compress() keeps the state the compressor was
    made with, multi_compress_to_buffer() sees a later precompute (setup_cctx runs once, the worker contexts are set up per call)"""
    pool = corpus.frame_list(40, 6)
    rng = np.random.default_rng(12)
    trained = ref.train_dictionary(16384, [f[j * 4096:(j + 1) * 4096] for f in pool for j in range(16)])
    rawd = pool[3][1000:9000]
    srcs = [(pool[0] + pool[1] + pool[2])[:n] for n in (1, 300, 4096, 8193, 16385, 60000, 131072, 131073, 250000)] + [rng.bytes(3000), (trained[-3000:] + pool[5])[:30000]]
    for dd, kw in ((trained, {}), (rawd, dict(dict_type=zstd.DICT_TYPE_RAWCONTENT))):
        for plevel, clevel in ((1, 3), (3, 1), (-3, 3), (2, 2)):
            d = zstd.ZstdCompressionDict(dd, **kw)
            assert d.precompute_compress(level=plevel) is None
            c = zstd.ZstdCompressor(level=clevel, dict_data=d)
            want = [ref.compress_with_cdict(s, dd, level=clevel, cdict_level=plevel, dict_type=kw.get("dict_type", 0)) for s in srcs]
            res = c.multi_compress_to_buffer(srcs)
            assert [res[i].tobytes() for i in range(len(srcs))] == want, (plevel, clevel)
            assert [c.compress(s) for s in srcs[2:6]] == want[2:6]
            # the frames are ordinary dictionary frames
            back = zstd.ZstdDecompressor(dict_data=zstd.ZstdCompressionDict(dd, **kw)).multi_decompress_to_buffer(want)
            assert [back[i].tobytes() for i in range(len(srcs))] == srcs
    # the level of the dictionary wins: these differ from the frames of a compressor that was simply given the level
    plain = [ref.compress(s, level=3, dict_data=trained) for s in srcs]
    assert plain != [ref.compress_with_cdict(s, trained, level=3, cdict_level=1) for s in srcs]
    # explicit parameters (to_cparams + ZSTD_createCDict_advanced); unset fields come from the default level's row
    for q in (dict(hash_log=10, chain_log=9, min_match=6, strategy=zstd.STRATEGY_FAST), dict(min_match=7, window_log=17), dict(chain_log=12)):
        d = zstd.ZstdCompressionDict(trained)
        d.precompute_compress(compression_params=zstd.ZstdCompressionParameters(**q))
        res = zstd.ZstdCompressor(level=3, dict_data=d).multi_compress_to_buffer(srcs)
        assert [res[i].tobytes() for i in range(len(srcs))] == [ref.compress_with_cdict(s, trained, level=3, cdict_params=q) for s in srcs], q
    # argument errors (tests/test_train_dictionary.py:77-93), a dictionary libzstd cannot digest (:95-107)
    d = zstd.ZstdCompressionDict(trained)
    with pytest.raises(ValueError, match="must specify one of level or "):
        d.precompute_compress()
    with pytest.raises(ValueError, match="must only specify one of level or "):
        d.precompute_compress(level=3, compression_params=zstd.ZstdCompressionParameters())
    zstd.ZstdCompressionDict(b"dictcontent" * 64, dict_type=zstd.DICT_TYPE_RAWCONTENT).precompute_compress(level=1)
    with pytest.raises(zstd.ZstdError, match="unable to precompute dictionary"):
        zstd.ZstdCompressionDict(b"dictcontent" * 64, dict_type=zstd.DICT_TYPE_FULLDICT).precompute_compress(level=1)
    # what the backend does not implement is refused here, loudly: strategies above double-fast, a window smaller than the dictionary
    with pytest.raises(zstd.ZstdError, match="unable to precompute dictionary"):
        zstd.ZstdCompressionDict(trained).precompute_compress(level=19)
    with pytest.raises(zstd.ZstdError, match="unable to precompute dictionary"):
        zstd.ZstdCompressionDict(trained).precompute_compress(compression_params=zstd.ZstdCompressionParameters(window_log=11))
    # order of construction: a compressor made BEFORE the precompute keeps its one-shot context, its batch workers see the CDict
    d = zstd.ZstdCompressionDict(trained)
    c = zstd.ZstdCompressor(level=3, dict_data=d)
    d.precompute_compress(level=1)
    assert c.compress(srcs[3]) == ref.compress(srcs[3], level=3, dict_data=trained)
    assert c.multi_compress_to_buffer([srcs[3]])[0].tobytes() == ref.compress_with_cdict(srcs[3], trained, level=3, cdict_level=1)
    # memory_size(): the reference reports its libzstd context (c-ext/compressor.c:263, decompressor.c:128); here the context lives on the
    # device -- after a call the calling thread's device arenas are what it reports
    assert c.memory_size() > 1000 and zstd.ZstdDecompressor().memory_size() == c.memory_size()
