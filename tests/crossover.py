"""Where the GPU path loses to the reference's CPU workers (measurement aid, test infrastructure; run on the GPU box).
For items of 4 KiB / 128 KiB / 1 MiB and batch sizes from 1 up: the wall time of ONE multi_compress_to_buffer / multi_decompress_to_buffer call through
Python on host buffers (PCIe and packing inclusive, best of 3 after a warm-up call) beside libzstd 1.5.7 on this host's threads over the same items
(oracle/zo_mtbench.c: one context per thread, static contiguous partition like the reference's batch workers, min(items, cores) threads and 64 threads,
the better of the two, best of 3 passes). Prints one JSON line per (item size, direction) with the per-batch times and the smallest measured batch from
which the GPU call is the faster one. Every GPU result of the small batches is compared with the input / libzstd's frames.
usage: python tests/crossover.py [quick]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zstandard_amd as pyz
from tests import reflib
from tests.corpus import Corpus
import bench

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
dev = torch.device("cuda", 0)
corpus = Corpus(device=dev)
ncpu = os.cpu_count() or 1
lib = bench._mtbench()


def cpu_ms(decompress, blob, offs, n, max_out, passes=3):
    best = None
    for t in sorted({min(n, ncpu), min(n, 64)}):
        times = (C.c_double * passes)()
        b = lib.zo_mt_bench_dict(reflib.REF_SO.encode(), 1 if decompress else 0, blob.ctypes.data, offs.ctypes.data, n, max_out, 3, t, passes, None, 0, times)
        assert b > 0
        if best is None or min(times) < best[0]:
            best = (min(times), t)
    return best[0] * 1e3, best[1]


def gpu_ms(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r


plans = [(4096, [1, 16, 128, 1024, 8192, 65536] + ([] if quick else [262144])),
         (131072, [1, 4, 16, 64, 256, 1024, 4096] + ([] if quick else [16384])),
         (1 << 20, [1, 4, 16, 64, 256, 1024] + ([] if quick else [2048]))]
cctx, dctx = pyz.ZstdCompressor(level=3), pyz.ZstdDecompressor()
ref = reflib.RefZstd()
for item, counts in plans:
    nmax = max(counts)
    per = max(1, item // 131072)
    if item >= 131072:
        raw = corpus.frames(0, nmax * per, chunk=256).reshape(nmax, item).cpu().numpy()
    else:
        raw = corpus.frames(0, (nmax * item + 131071) // 131072, chunk=256).reshape(-1)[: nmax * item].reshape(nmax, item).cpu().numpy()
    frames, csizes = bench.compress_on_host(raw, item)
    rows = {"compress": [], "decompress": []}
    for n in counts:
        items = [raw[i].tobytes() for i in range(n)] if n <= 4096 else None
        segs = np.zeros((n, 2), dtype=np.uint64); segs[:, 0] = np.arange(n, dtype=np.uint64) * item; segs[:, 1] = item
        src = pyz.BufferWithSegments(memoryview(raw[:n]).cast("B"), segs.tobytes())
        g_c, r = gpu_ms(lambda: cctx.multi_compress_to_buffer(src))
        for i in (0, n // 2, n - 1):
            assert r[i].tobytes() == frames[i], "GPU frame %d of %d x %d differs from libzstd" % (i, n, item)
        del r
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(item)
        c_c, t_c = cpu_ms(False, np.ascontiguousarray(raw[:n]), offs, n, 0)
        blob = np.frombuffer(b"".join(frames[:n]), dtype=np.uint8)
        foffs = np.zeros(n + 1, dtype=np.uint64); foffs[1:] = np.cumsum(csizes[:n])
        fsegs = np.zeros((n, 2), dtype=np.uint64); fsegs[:, 0] = foffs[:n]; fsegs[:, 1] = np.asarray(csizes[:n], dtype=np.uint64)
        fb = pyz.BufferWithSegments(memoryview(blob), fsegs.tobytes())
        sizes = np.full(n, item, dtype=np.uint64).tobytes()
        g_d, r = gpu_ms(lambda: dctx.multi_decompress_to_buffer(fb, decompressed_sizes=sizes))
        for i in (0, n // 2, n - 1):
            assert r[i].tobytes() == raw[i].tobytes(), "GPU output %d of %d x %d differs from the input" % (i, n, item)
        del r
        c_d, t_d = cpu_ms(True, blob, foffs, n, item)
        rows["compress"].append({"n": n, "gpu_ms": round(g_c, 3), "cpu_ms": round(c_c, 3), "cpu_threads": t_c})
        rows["decompress"].append({"n": n, "gpu_ms": round(g_d, 3), "cpu_ms": round(c_d, 3), "cpu_threads": t_d})
    for direction, rr in rows.items():
        cross = None
        for k in range(len(rr)):
            if all(x["gpu_ms"] < x["cpu_ms"] for x in rr[k:]):
                cross = rr[k]["n"]; break
        print(json.dumps({"item_bytes": item, "direction": direction, "host_cores": ncpu, "gpu_faster_from_n": cross, "rows": rr}), flush=True)
    del raw, frames
