"""Batches of SMALL sources (no dictionary) through the host-buffer API: the LDS-source match kernel with its LDS area sized for the
batch's largest source (more frames per CU) against the flat kernel (ZHIP_E1LDS_MAX=0). ms per multi_compress_to_buffer call, a sample
of the frames checked against libzstd 1.5.7.   Usage: python tests/small_source_batches.py   (once per ZHIP_E1LDS_MAX setting)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import zstandard_amd as pyz
from tests import reflib
from tests.corpus import Corpus

corpus = Corpus(device=torch.device("cuda", 0))
ref = reflib.RefZstd()
raw = corpus.frames(0, 2048, chunk=256).cpu().numpy()
c = pyz.ZstdCompressor(level=3)
out = {"ZHIP_E1LDS_MAX": os.environ.get("ZHIP_E1LDS_MAX", "default")}
for size, counts in ((4096, (1024, 8192, 16384, 32768)), (16384, (1024, 4608, 9216)), (65536, (512, 1024, 2048))):
    per = 131072 // size
    items = [raw[i // per][(i % per) * size:(i % per + 1) * size].tobytes() for i in range(max(counts))]
    want = [ref.compress(x) for x in items[:48]]
    for n in counts:
        c.multi_compress_to_buffer(items[:n])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = c.multi_compress_to_buffer(items[:n]); best = min(best, time.perf_counter() - t0)
        assert all(r[i].tobytes() == want[i] for i in range(48)) and len(r) == n
        out["%dB_x%d_ms" % (size, n)] = round(best * 1e3, 2)
print(json.dumps(out))
