"""PCIe-inclusive rate of the host-buffer API -- the Python-visible multi_compress_to_buffer / multi_decompress_to_buffer calls through
zhip_*_batch: host packing + H2D + kernels + D2H, pipelined over three streams -- next to the reference libzstd on the host's own
threads over the same frames (what a python-zstandard user would get from the reference's multi_*_to_buffer(threads=-1)).
bench.py's `value` is the HBM-resident rate; this is the number for DESIGN.md section 3.   Usage: python tests/host_api_rate.py [frames]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (same HIP runtime instance as the library)
import zstandard_amd as pyz
from tests.corpus import Corpus
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
FRAME = 131072
corpus = Corpus(device=torch.device("cuda", 0))
raw = corpus.frames(0, F, chunk=256).cpu().numpy()
frames, csizes = bench.compress_on_host(raw, FRAME)
blob = b"".join(frames)
segs = np.zeros((F, 2), dtype=np.uint64)
segs[:, 1] = csizes
segs[1:, 0] = np.cumsum(segs[:-1, 1])
out = {"frames": F}
bws = pyz.BufferWithSegments(blob, segs.tobytes())
sizes = np.full(F, FRAME, dtype=np.uint64).tobytes()
d = pyz.ZstdDecompressor()
r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes)                      # warm-up: device arenas, pinned staging, pinned payload pool
del r
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); t1 = time.perf_counter()
    best = min(best, t1 - t0)
    assert len(r) == F and r[F - 1].tobytes() == raw[F - 1].tobytes() and r[F // 2 + 1].tobytes() == raw[F // 2 + 1].tobytes() and r[0].tobytes() == raw[0].tobytes()
    del r
out["decompress_GBps"] = round(F * FRAME / best / 1e9, 2)
rsegs = np.zeros((F, 2), dtype=np.uint64); rsegs[:, 0] = np.arange(F, dtype=np.uint64) * FRAME; rsegs[:, 1] = FRAME
rb = pyz.BufferWithSegments(raw.tobytes(), rsegs.tobytes())
c = pyz.ZstdCompressor(level=3)
r = c.multi_compress_to_buffer(rb)
del r
best = 1e9
for _ in range(2):
    t0 = time.perf_counter(); r = c.multi_compress_to_buffer(rb); t1 = time.perf_counter()
    best = min(best, t1 - t0)
    assert len(r) == F and r[5].tobytes() == frames[5] and r[F - 1].tobytes() == frames[F - 1] and r[F // 2 + 1].tobytes() == frames[F // 2 + 1]
    del r
out["compress_GBps"] = round(F * FRAME / best / 1e9, 2)
# one-shot calls (the latency a single .compress() / .decompress() pays)
one = raw[3].tobytes()
c.compress(one); t0 = time.perf_counter(); f1 = c.compress(one); out["one_shot_compress_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
assert f1 == frames[3]
d.decompress(f1); t0 = time.perf_counter(); b1 = d.decompress(f1); out["one_shot_decompress_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
assert b1 == one
# the reference on this host's threads over the same data (native threads, oracle/zo_mtbench.c)
ns = min(F, 16384)
bl, bo = bench.sample_blob(frames, ns)
cd = bench.cpu_baseline(True, bl, bo, ns, FRAME, ns * FRAME)
offs = np.arange(ns + 1, dtype=np.uint64) * np.uint64(FRAME)
cc = bench.cpu_baseline(False, np.ascontiguousarray(raw[:ns]), offs, ns, 0, ns * FRAME)
out["libzstd_host_decompress_GBps"] = {"median": cd["value"], "threads": cd["cores"], "by_threads": cd["by_threads"]}
out["libzstd_host_compress_GBps"] = {"median": cc["value"], "threads": cc["cores"], "by_threads": cc["by_threads"]}
print("host-API (uncompressed GB/s, PCIe + packing inclusive, %d x 128 KiB): " % F + json.dumps(out))
