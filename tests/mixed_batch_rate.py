"""A batch of mostly small frames with a few large ones through the host-buffer API (PCIe inclusive): python tests/mixed_batch_rate.py [small] [large] [MiB].
The large frames put the decode kernels in their several-block mode; the pool of block slots is sized from ALL the frames' sizes, so the small
ones are not cut into chunks sized for the large ones."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zstandard_amd as pyz
from tests import reflib
from tests.corpus import Corpus

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
MIB = int(sys.argv[3]) if len(sys.argv) > 3 else 4
corpus = Corpus(device=torch.device("cuda", 0)); ref = reflib.RefZstd()
raw = corpus.frames(0, 2048, chunk=256).cpu().numpy()
small = [raw[i % 2048].tobytes() for i in range(S)]
big = [b"".join(raw[(8 * MIB * j + k) % 2048].tobytes() for k in range(8 * MIB)) for j in range(B)]
fs = {}
for i in range(2048): fs[i] = ref.compress(raw[i].tobytes())
frames = [fs[i % 2048] for i in range(S)]
for j, b in enumerate(big): frames.insert((j + 1) * (S // (B + 1)), ref.compress(b))
raws = list(small)
for j, b in enumerate(big): raws.insert((j + 1) * (S // (B + 1)), b)
d = pyz.ZstdDecompressor()
out = {"small_frames": S, "large_frames": B, "large_MiB": MIB}
d.multi_decompress_to_buffer(frames)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(frames); best = min(best, time.perf_counter() - t0)
assert all(r[i].tobytes() == raws[i] for i in range(0, len(raws), 97)) and all(r[(j + 1) * (S // (B + 1))].tobytes() == big[j] for j in range(B))
out["decompress_ms"] = round(best * 1e3, 1); out["decompress_GBps"] = round(sum(map(len, raws)) / best / 1e9, 2)
print(json.dumps(out))
