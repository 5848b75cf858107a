"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/zstd_hip.h declares,
the host-side mirror validates arguments like the reference (no compute calls here: there is no GPU and no fallback)."""
import os
import re
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ss = struct.Struct("=QQ")


@pytest.fixture(scope="module")
def zstd():
    import zstandard_amd
    return zstandard_amd


def test_library_exports_every_declared_symbol(zstd):
    header = open(os.path.join(ROOT, "include", "zstd_hip.h")).read()
    declared = set(re.findall(r"\b(zhip_[a-z0-9_]+)\s*\(", header))
    lib = zstd._lib.lib()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libzstd_hip.so does not export %s" % name
    assert set(zstd._lib.EXPORTED_SYMBOLS) <= declared
    assert lib.zhip_abi_version() == 3
    assert lib.zhip_compress_bound(131072) == 131072 + 512
    assert lib.zhip_compress_bound(0) == 64


def test_frame_inspection_entry_points(zstd):
    import ctypes as C
    lib = zstd._lib.lib()
    f = bytes.fromhex("28b52ffd2000010000")
    buf = C.create_string_buffer(f, len(f))
    assert lib.zhip_frame_content_size(C.cast(buf, C.c_void_p), len(f)) == 0
    assert lib.zhip_find_frame_compressed_size(C.cast(buf, C.c_void_p), len(f)) == len(f)
    bad = C.create_string_buffer(b"foobarbaz", 9)
    assert lib.zhip_frame_content_size(C.cast(bad, C.c_void_p), 9) == zstd._lib.CONTENTSIZE_ERROR
    assert lib.zhip_error_name(20) == b"Data corruption detected"
    assert lib.zhip_error_name(70) == b"Destination buffer is too small"


def test_backend_surface(zstd):
    assert zstd.backend_features >= {"buffer_types", "multi_compress_to_buffer", "multi_decompress_to_buffer"}
    for name in ("ZstdCompressor", "ZstdDecompressor", "BufferWithSegments", "BufferWithSegmentsCollection", "BufferSegment",
                 "BufferSegments", "ZstdCompressionDict", "ZstdError"):
        assert hasattr(zstd, name)


# ---- mirrors of the reference's tests/test_buffer_util.py
def test_buffer_with_segments(zstd):
    with pytest.raises(TypeError):
        zstd.BufferWithSegments()
    with pytest.raises(TypeError):
        zstd.BufferWithSegments(b"foo")
    with pytest.raises(ValueError, match="segments array size is not a multiple of 16"):
        zstd.BufferWithSegments(b"foo", b"\x00\x00")
    with pytest.raises(ValueError, match="offset within segments array references memory"):
        zstd.BufferWithSegments(b"foo", ss.pack(0, 4))
    b = zstd.BufferWithSegments(b"foo", ss.pack(0, 3))
    with pytest.raises(IndexError, match="offset must be non-negative"):
        b[-10]
    with pytest.raises(IndexError, match="offset must be less than 1"):
        b[1]
    assert len(b) == 1 and b.size == 3 and b.tobytes() == b"foo"
    assert len(b[0]) == 3 and b[0].offset == 0 and b[0].tobytes() == b"foo"
    b = zstd.BufferWithSegments(b"foofooxfooxy", b"".join([ss.pack(0, 3), ss.pack(3, 4), ss.pack(7, 5)]))
    assert len(b) == 3 and b.size == 12
    assert [b[i].tobytes() for i in range(3)] == [b"foo", b"foox", b"fooxy"]
    assert b.segments().tobytes() == b"".join([ss.pack(0, 3), ss.pack(3, 4), ss.pack(7, 5)])


def test_buffer_with_segments_collection(zstd):
    with pytest.raises(ValueError, match="must pass at least 1 argument"):
        zstd.BufferWithSegmentsCollection()
    with pytest.raises(TypeError, match="arguments must be BufferWithSegments"):
        zstd.BufferWithSegmentsCollection(None)
    with pytest.raises(ValueError, match="ZstdBufferWithSegments cannot be empty"):
        zstd.BufferWithSegmentsCollection(zstd.BufferWithSegments(b"", b""))
    b1 = zstd.BufferWithSegments(b"foo", ss.pack(0, 3))
    b2 = zstd.BufferWithSegments(b"barbaz", b"".join([ss.pack(0, 3), ss.pack(3, 3)]))
    c = zstd.BufferWithSegmentsCollection(b1, b2)
    assert len(c) == 3 and c.size() == 9
    with pytest.raises(IndexError, match="offset must be less than 3"):
        c[3]
    assert [c[i].tobytes() for i in range(3)] == [b"foo", b"bar", b"baz"]


# ---- argument validation of the hot-path entry points (reference: compressor.c:88-246, 1340-1503; decompressor.c:1459-1710)
def test_compressor_argument_validation(zstd):
    with pytest.raises(ValueError, match="level must be less than 23"):
        zstd.ZstdCompressor(level=23)
    with pytest.raises(TypeError):
        zstd.ZstdCompressor(dict_data=b"raw bytes")
    c = zstd.ZstdCompressor()
    with pytest.raises(TypeError, match="argument must be list of BufferWithSegments"):
        c.multi_compress_to_buffer(True)
    with pytest.raises(ValueError, match="no source elements found"):
        c.multi_compress_to_buffer([])
    with pytest.raises(ValueError, match="source elements are empty"):
        c.multi_compress_to_buffer([b"", b""])
    with pytest.raises(TypeError, match="item 1 not a bytes like object"):
        c.multi_compress_to_buffer([b"ok", 7])


def test_decompressor_argument_validation(zstd):
    d = zstd.ZstdDecompressor()
    with pytest.raises(TypeError):
        d.multi_decompress_to_buffer(True)
    with pytest.raises(TypeError):
        d.multi_decompress_to_buffer((1, 2))
    with pytest.raises(TypeError, match="item 0 not a bytes like object"):
        d.multi_decompress_to_buffer(["foo"])
    with pytest.raises(ValueError, match="decompressed_sizes size mismatch; expected 16, got 8"):
        d.multi_decompress_to_buffer([b"a", b"b"], decompressed_sizes=struct.pack("=Q", 1))
    with pytest.raises(zstd.ZstdError, match="read_across_frames=True is not yet implemented"):
        d.decompress(b"whatever", read_across_frames=True)
    with pytest.raises(zstd.ZstdError, match="error determining content size from frame header"):
        d.decompress(b"")
    assert zstd.ZstdCompressionDict(b"\x37\xa4\x30\xec" + struct.pack("<I", 1234) + b"x" * 100).dict_id() == 1234
    assert zstd.ZstdCompressionDict(b"plain content").dict_id() == 0


def test_partition_by_bytes_matches_reference_rule():
    import importlib
    par = importlib.import_module("python-zstandard_amd.parallel")
    sizes = [10] * 10
    assert par.partition_by_bytes(sizes, 2) == [(0, 5), (5, 10)]
    assert par.partition_by_bytes(sizes, 3) == [(0, 4), (4, 8), (8, 10)]          # cut once a worker reaches total/workers
    assert par.partition_by_bytes([100, 1, 1, 1], 2) == [(0, 1), (1, 4)]
    # fewer items than workers: the reference clamps the worker count (compressor.c:1151); here every ASKED worker (= rank) still gets an
    # entry, empty beyond the items, because a rank without work must take part in the collectives (IndexError + hang)
    assert par.partition_by_bytes([5], 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    assert par.partition_by_bytes([10, 20, 30], 8) == [(0, 2), (2, 3)] + [(3, 3)] * 6
    assert par.partition_by_bytes([], 4) == [(0, 0)] * 4
    bounds = par.partition_by_bytes(list(range(1, 100)), 8)
    assert bounds[0][0] == 0 and bounds[-1][1] == 99 and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))


def test_native_cpu_baseline_driver_runs(ref, corpus):
    """oracle/libzo_mtbench.so (bench.py's cpu_baseline leg): both directions over a few frames with two threads; the frames it
    compresses are the reference's, so it must at least succeed and report a positive time"""
    import ctypes as C
    import os
    import subprocess
    import numpy as np
    from tests import reflib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "libzo_mtbench.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "libzo_mtbench.so"])
    lib = C.CDLL(so)
    lib.zo_mt_bench.restype = C.c_double
    lib.zo_mt_bench.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_int, C.c_int, C.c_int]
    raws = [corpus.frame_bytes(i) for i in range(6)]
    raw = np.frombuffer(b"".join(raws), dtype=np.uint8)
    offs = (np.arange(7, dtype=np.uint64) * np.uint64(131072))
    t = lib.zo_mt_bench(reflib.REF_SO.encode(), 0, raw.ctypes.data, offs.ctypes.data, 6, 0, 3, 2, 1)
    assert t > 0
    frames = [ref.compress(r) for r in raws]
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    foffs = np.zeros(7, dtype=np.uint64); foffs[1:] = np.cumsum([len(f) for f in frames])
    t = lib.zo_mt_bench(reflib.REF_SO.encode(), 1, blob.ctypes.data, foffs.ctypes.data, 6, 131072, 3, 2, 1)
    assert t > 0
    # a damaged frame makes the pass fail loudly instead of timing garbage
    bad = bytearray(blob.tobytes()); bad[int(foffs[2]) + 20] ^= 0xFF
    badnp = np.frombuffer(bytes(bad), dtype=np.uint8)
    assert lib.zo_mt_bench(reflib.REF_SO.encode(), 1, badnp.ctypes.data, foffs.ctypes.data, 6, 131072, 3, 2, 1) < 0


def test_integration_stub_compiles_against_the_header(tmp_path):
    """INTEGRATION.md's reference-side binding (struct literals, calls, ownership hand-off) as C, compiled against include/zstd_hip.h"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(root, "tests", "abi_stub_check.c"), "-o", str(tmp_path / "abi_stub_check.o")])


def test_bench_line_shape_for_the_driver():
    """bench.py's roofline object keeps every key round-to-round comparisons read (a mid-line comment once swallowed
    `kernel_ms_per_step` / `algorithmic_bytes_per_step`), and the printed line ends with the flat `summary` -- the driver keeps the TAIL of
    stdout, so the compress / combined / host-API figures must be the last two KB of the line."""
    import json
    import bench

    class FakeCtx:
        def kernel_name(self, k):
            return ["zhip_decode_lit_kernel", "zhip_decode_exec_kernel"][k]

    rl, kdom = bench.roofline(FakeCtx(), {0: (2.0, 4), 1: (10.0, 4)}, 4, 11_000_000_000, 12.5, 65536)
    assert kdom == 1
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_step", "algorithmic_bytes_per_step", "dominant_kernel", "end_to_end"):
        assert k in rl, k
    assert rl["kernel_ms_per_step"] == 12.0 and rl["algorithmic_bytes_per_step"] == 11_000_000_000

    class SpanCtx:          # K1b beside K2: the kernels' durations overlap, the library's span of the chunk's pipeline is the step's kernel time
        def kernel_name(self, k):
            return ["zhip_decode_frames_kernel", "zhip_decode_huf_kernel", "zhip_decode_seq_kernel", "zhip_decode_exec_kernel", bench.SPAN_NAME][k]

    kt = {0: (0.5, 4), 1: (2.4, 4), 2: (8.5, 4), 3: (10.0, 4), 4: (21.0, 4)}
    rl2, kdom2 = bench.roofline(SpanCtx(), kt, 4, 11_000_000_000, 22.0, 65536)
    assert kdom2 == 3 and rl2["kernel_ms_per_step"] == 21.5 and bench.SPAN_NAME not in rl2["kernel"]
    ko = bench.kernels_obj(SpanCtx(), kt)
    assert "note" in ko[bench.SPAN_NAME] and "note" in ko["zhip_decode_huf_kernel"]
    line = {"metric": "m", "value": 350.0, "unit": "GB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 24.4, "config": {"workload": "w"},
            "roofline": rl, "cpu_baseline": {"value": 25.0}, "kernels": {"x": {"pad": "y" * 4000}},
            "compress": {"value": 19.5, "ms_per_step": 440.0, "roofline": {"frac": 0.0033}, "regime": {"class": "fast", "match_kernel_ms_per_65536_frames": 405.7},
                         "kernels": {"pad": "z" * 4000}},
            "combined": {"value": 18.5}, "host_api": {"devices": 1, "frames_8192": {"compress": 6.4, "decompress": 28.6}, "frames_65536": {"compress": 14.0, "decompress": 44.8}},
            "dict": {"value": 26.0, "decompress": {"value": 130.0}}, "roundtrip": {"value": 19.7, "compress": {"value": 20.8}, "decompress": {"value": 358.0}},
            "blocks": {"value": 152.0, "compress": {"value": 2.1}}, "verified": True}
    out = bench.ordered_for_the_driver(line)
    assert set(out) == set(line) | {"summary"} and all(out[k] == line[k] for k in line)
    assert list(out)[-2:] == ["summary", "verified"]
    tail = json.dumps(out)[-2000:]
    for k in ("compress_gbs", "combined_gbs", "host_api_65536_c", "host_api_8192_c", "dict_d_gbs", "roundtrip_c_gbs", "decompress_gbs", "compress_match_kernel_ms"):
        assert '"%s"' % k in tail, k


def test_c_partition_rule_is_the_references(zstd):
    """zhip_partition_by_bytes (what zhip_compress_batch / zhip_decompress_batch cut a batch over the node's devices with) against the reference's
    dispatcher loop restated here line by line (c-ext/compressor.c:1127-1216: bytesPerWorker = total / threadCount, a worker's run closes once its
    bytes reach it, the last worker takes what is left, never more workers than sources :1151) and against parallel.py's form of the same rule."""
    import ctypes as C
    import importlib
    import random
    par = importlib.import_module("python-zstandard_amd.parallel")
    lib = zstd._lib.lib()
    lib.zhip_partition_by_bytes.restype = C.c_size_t
    lib.zhip_partition_by_bytes.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]

    def reference_rule(sizes, threads):
        threads = max(1, min(threads, len(sizes)))
        per = sum(sizes) // threads
        runs, cur, start, acc = [], 0, 0, 0
        for i, s in enumerate(sizes):
            acc += s
            if cur == threads - 1:
                continue
            if acc >= per:
                runs.append((start, i + 1)); cur += 1; start = i + 1; acc = 0
        if acc:
            runs.append((start, len(sizes)))
        return runs

    def c_rule(sizes, workers):
        arr = (C.c_uint64 * max(1, len(sizes)))(*sizes)
        out = (C.c_size_t * (2 * max(1, workers)))()
        used = lib.zhip_partition_by_bytes(arr, len(sizes), workers, out)
        return [(out[2 * w], out[2 * w + 1]) for w in range(used)]

    rng = random.Random(6)
    cases = [([10] * 10, 2), ([10] * 10, 3), ([100, 1, 1, 1], 2), ([5], 4), ([10, 20, 30], 8), (list(range(1, 100)), 8), ([131072] * 65536, 8), ([], 4)]
    for _ in range(300):
        n = rng.randint(1, 60)
        cases.append(([rng.choice([1, 7, 4096, 131072, rng.randint(1, 10 ** 6)]) for _ in range(n)], rng.randint(1, 9)))
    for sizes, w in cases:
        got = c_rule(sizes, w)
        assert got == reference_rule(sizes, w), (sizes[:8], w)
        assert got == [b for b in par.partition_by_bytes(sizes, w) if b[1] > b[0]], (sizes[:8], w)
        if sizes:
            assert got[0][0] == 0 and got[-1][1] == len(sizes) and all(a[1] == b[0] for a, b in zip(got, got[1:]))
