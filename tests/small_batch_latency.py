"""Latency of small compress batches through the host-buffer API (one-shot .compress() is the batch of one): the LDS-source match
kernel (ze_match_lds_body, the default up to four frames per CU) against the flat kernel (ZHIP_E1LDS_MAX=0), every frame checked
against libzstd 1.5.7.   Usage: python tests/small_batch_latency.py   (run once per setting of ZHIP_E1LDS_MAX)"""
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
raw = corpus.frames(0, 1024, chunk=256).cpu().numpy()
items = [raw[i].tobytes() for i in range(1024)]
want = [ref.compress(x) for x in items[:64]]
c = pyz.ZstdCompressor(level=3)
out = {"ZHIP_E1LDS_MAX": os.environ.get("ZHIP_E1LDS_MAX", "default")}
c.compress(items[0])
ts = []
for k in range(5):
    t0 = time.perf_counter(); f = c.compress(items[k]); ts.append(time.perf_counter() - t0)
    assert f == want[k]
out["one_shot_128KiB_ms"] = round(min(ts) * 1e3, 2)
for size in (4096, 32768):
    x = items[3][:size]; c.compress(x)
    t0 = time.perf_counter(); f = c.compress(x); out["one_shot_%d_ms" % size] = round((time.perf_counter() - t0) * 1e3, 2)
    assert f == ref.compress(x)
for n in (16, 128, 256, 512, 1024):
    c.multi_compress_to_buffer(items[:n])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = c.multi_compress_to_buffer(items[:n]); best = min(best, time.perf_counter() - t0)
    assert all(r[i].tobytes() == want[i] for i in range(min(n, 64)))
    out["batch_%d_ms" % n] = round(best * 1e3, 2)
d = pyz.ZstdDecompressor()
frames = [ref.compress(x) for x in items[:256]]
d.decompress(frames[0])
ts = []
for k in range(5):
    t0 = time.perf_counter(); b = d.decompress(frames[k]); ts.append(time.perf_counter() - t0)
    assert b == items[k]
out["one_shot_decompress_128KiB_ms"] = round(min(ts) * 1e3, 2)
for n in (16, 256):
    d.multi_decompress_to_buffer(frames[:n])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(frames[:n]); best = min(best, time.perf_counter() - t0)
    assert all(r[i].tobytes() == items[i] for i in range(n))
    out["decompress_batch_%d_ms" % n] = round(best * 1e3, 2)
# one large frame (several blocks): the several-block mode of the decode kernels
for mib in (1, 8):
    big = b"".join(items[: 8 * mib])
    fb = ref.compress(big)
    d.decompress(fb)
    ts = []
    for k in range(3):
        t0 = time.perf_counter(); b = d.decompress(fb); ts.append(time.perf_counter() - t0)
    assert b == big
    out["one_shot_decompress_%dMiB_ms" % mib] = round(min(ts) * 1e3, 2)
    ts = []
    c.compress(big)
    for k in range(2):
        t0 = time.perf_counter(); f = c.compress(big); ts.append(time.perf_counter() - t0)
    assert f == fb
    out["one_shot_compress_%dMiB_ms" % mib] = round(min(ts) * 1e3, 2)
print(json.dumps(out))
