"""GPU stress of the single-block compress path through the Python API against libzstd 1.5.7: python tests/stress_gpu_compress.py SEED [SOURCES].
Sources of 1 byte ... 128 KiB of the kinds the emulator's search tests use (corpus slices, random, few-symbol, periodic, damaged corpus, zeros, runs of
random length), one multi_compress_to_buffer call per setting: the default kernels (the LDS-source kernel for small batches, the flat kernel with four /
three / two probes per trip by launch size) and ZHIP_E1LDS_MAX=0 (every batch through the flat kernel); level 3 and a fast-strategy level. Every frame
must be libzstd's. The emulator twin is tests/stress_emu_encode.py. Not collected by pytest."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstandard_amd as zstd
from tests import reflib
from tests.corpus import Corpus

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ref = reflib.RefZstd(); corpus = Corpus(); rng = np.random.default_rng(seed)
raws = []
for i in range(count):
    kind = i % 8
    n = int(rng.integers(1, 131073)) if i % 3 else int(rng.integers(1, 5000))
    if kind == 0: r = corpus.frame_bytes(int(rng.integers(0, 1000)))[:n]
    elif kind == 1: r = rng.bytes(n)
    elif kind == 2: r = bytes(rng.integers(0, 3, n, dtype=np.uint8))
    elif kind == 3: r = (b"abcdefgh" * (n // 8 + 1))[:n]
    elif kind == 4: r = (rng.bytes(int(rng.integers(1, 900))) * (n + 1))[:n]
    elif kind == 5:
        a = bytearray(corpus.frame_bytes(int(rng.integers(0, 1000)))[:n])
        for k in range(0, len(a), 997): a[k] = int(rng.integers(0, 256))
        r = bytes(a)
    elif kind == 6: r = b"\0" * n
    else:
        parts, tot = [], 0
        while tot < n:
            m = int(rng.integers(1, 5000))
            parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0, 256))]) * m); tot += m
        r = b"".join(parts)[:n]
    raws.append(r)
want = {lvl: [ref.compress(r, level=lvl) for r in raws] for lvl in (3, 1)}
bad = 0
t0 = time.time()
for env in ({}, {"ZHIP_E1LDS_MAX": "0"}):
    box = {}

    def run():
        try:
            for lvl in (3, 1):
                res = zstd.ZstdCompressor(level=lvl).multi_compress_to_buffer(raws)
                box[lvl] = [res[i].tobytes() for i in range(len(raws))]
            sub = raws[:200]
            res = zstd.ZstdCompressor(level=3).multi_compress_to_buffer(sub)          # a small batch: the LDS-source kernel where it is on
            box["small"] = [res[i].tobytes() for i in range(len(sub))]
        except Exception as e:              # noqa: BLE001
            box["error"] = e

    os.environ.update(env)
    try:
        t = threading.Thread(target=run); t.start(); t.join()        # (a fresh thread: the knobs are read when its context is created)
    finally:
        for k in env: del os.environ[k]
    if "error" in box:
        print("ERROR", env, box["error"]); bad += 1; continue
    for lvl in (3, 1):
        for i, (o, w) in enumerate(zip(box[lvl], want[lvl])):
            if o != w:
                bad += 1; print("MISMATCH", env, "level", lvl, i, len(raws[i]))
    for i, (o, w) in enumerate(zip(box["small"], want[3][:200])):
        if o != w:
            bad += 1; print("MISMATCH small", env, i, len(raws[i]))
print("gpu compress stress seed %d: %d sources x 2 levels x 2 settings, bad %d, %.1fs" % (seed, count, bad, time.time() - t0))
