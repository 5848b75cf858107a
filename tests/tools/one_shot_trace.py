"""Where a one-shot .compress() / .decompress() of one 128 KiB source spends its time: wall clock of the call through the Python API (five calls each after a warm-up),
to be run plain and under `rocprofv3 --kernel-trace --hip-trace --stats` (kernel durations and HIP API calls of the same process).  Usage: python tests/tools/one_shot_trace.py [bytes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import zstandard_amd as pyz
from tests.corpus import Corpus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
raw = Corpus(device=torch.device("cuda", 0), frame_size=n).frames(0, 8, chunk=8).cpu().numpy()
one = raw[3].tobytes()
c = pyz.ZstdCompressor(level=3)
d = pyz.ZstdDecompressor()
f = c.compress(one); c.compress(one)
assert d.decompress(f) == one; d.decompress(f)
tc, td = [], []
for _ in range(5):
    t0 = time.perf_counter(); f1 = c.compress(one); tc.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); b1 = d.decompress(f1); td.append((time.perf_counter() - t0) * 1e3)
    assert b1 == one
print("one-shot %d bytes: compress ms %s  decompress ms %s" % (n, [round(x, 2) for x in tc], [round(x, 2) for x in td]))
