"""The host-buffer decompress call against the size of its pipeline's FIRST chunk (ZHIP_HCHUNK_D0; A/B aid, test infrastructure; run on the GPU box).
65 536 x 128 KiB frames through ZstdDecompressor.multi_decompress_to_buffer on host buffers, best of 3 calls after a warm-up call, one fresh thread per setting
(the knob is read when a thread's context is created). Also a plain pinned device-to-host copy of 1 GiB pieces: what the link gives."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import zstandard_amd as pyz
from tests.corpus import Corpus
import bench

F, FRAME = 65536, 131072
dev = torch.device("cuda", 0)
raw = Corpus(device=dev).frames(0, F, chunk=256).cpu().numpy()
frames, csizes = bench.compress_on_host(raw, FRAME)
blob = b"".join(frames)
segs = np.zeros((F, 2), dtype=np.uint64); segs[:, 1] = csizes; segs[1:, 0] = np.cumsum(segs[:-1, 1])
bws = pyz.BufferWithSegments(blob, segs.tobytes())
sizes = np.full(F, FRAME, dtype=np.uint64).tobytes()
g = torch.empty(1 << 30, dtype=torch.uint8, device=dev); h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
h.copy_(g); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    h.copy_(g, non_blocking=True)
torch.cuda.synchronize()
print("pinned D2H of 8 x 1 GiB: %.1f GB/s" % (8 * (1 << 30) / (time.perf_counter() - t0) / 1e9), flush=True)
t0 = time.perf_counter()
for _ in range(8):
    g.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print("pinned H2D of 8 x 1 GiB: %.1f GB/s" % (8 * (1 << 30) / (time.perf_counter() - t0) / 1e9), flush=True)
del g, h


def run(tag):
    d = pyz.ZstdDecompressor()
    r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); del r
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); t1 = time.perf_counter()
        best = min(best, t1 - t0)
        assert len(r) == F and r[F - 1].tobytes() == raw[F - 1].tobytes() and r[0].tobytes() == raw[0].tobytes() and r[4097].tobytes() == raw[4097].tobytes()
        del r
    print("ZHIP_HCHUNK_D0=%-6s decompress %.2f GB/s (%.1f ms)" % (tag, F * FRAME / best / 1e9, best * 1e3), flush=True)


for v in (sys.argv[1:] or ["0", "1024", "2048", "4096", "0", "2048"]):
    os.environ["ZHIP_HCHUNK_D0"] = v
    t = threading.Thread(target=run, args=(v,)); t.start(); t.join()
