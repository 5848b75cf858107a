"""PCIe-inclusive rate of the host-buffer API (multi_*_to_buffer through zhip_*_batch: staging copy + H2D + kernels + D2H), for the
note in DESIGN.md; bench.py's `value` is the HBM-resident rate. Usage: python tests/host_api_rate.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (same HIP runtime instance as the library)
import zstandard_amd as pyz
from tests.corpus import Corpus
from tests import reflib
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
corpus = Corpus(device=torch.device("cuda", 0))
raw = corpus.frames(0, F, chunk=256).cpu().numpy()
ref = reflib.RefZstd()
frames = bench.compress_on_host(ref, raw, 64)
blob = b"".join(frames)
segs = np.zeros((F, 2), dtype=np.uint64)
segs[:, 1] = [len(f) for f in frames]
segs[1:, 0] = np.cumsum(segs[:-1, 1])
out = {}
for name, mod in (("cext", pyz.load_cext()), ("python", pyz)):
    bws = mod.BufferWithSegments(blob, segs.tobytes())
    sizes = np.full(F, 131072, dtype=np.uint64).tobytes()
    d = mod.ZstdDecompressor()
    d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes)                      # warm-up: allocations
    t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); t1 = time.perf_counter()
    assert len(r) == F and r[F - 1].tobytes() == raw[F - 1].tobytes()
    out["decompress_" + name] = F * 131072 / (t1 - t0) / 1e9
    rsegs = np.zeros((F, 2), dtype=np.uint64); rsegs[:, 0] = np.arange(F, dtype=np.uint64) * 131072; rsegs[:, 1] = 131072
    rb = mod.BufferWithSegments(raw.tobytes(), rsegs.tobytes())
    c = mod.ZstdCompressor(level=3)
    c.multi_compress_to_buffer(rb)
    t0 = time.perf_counter(); r = c.multi_compress_to_buffer(rb); t1 = time.perf_counter()
    assert r[5].tobytes() == frames[5]
    out["compress_" + name] = F * 131072 / (t1 - t0) / 1e9
print("host-API GB/s (uncompressed bytes, PCIe + staging inclusive, %d x 128 KiB): " % F + ", ".join("%s %.2f" % kv for kv in sorted(out.items())))
