"""Emulator stress of the decode pipeline against libzstd 1.5.7 frames: python tests/stress_emu_decode.py SEED.
Not collected by pytest; the bounded version lives in test_emu_kernels.py."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus
emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
raws = []
for i in range(24):
    kind = i % 8
    n = int(rng.integers(1000, 131073))
    if kind in (0, 1, 2): r = corpus.frame_bytes(int(rng.integers(0, 2000)))[:n]
    elif kind == 3: r = rng.bytes(n)
    elif kind == 4:
        blk = rng.bytes(700); r = (blk + rng.bytes(3000) + blk * 5 + rng.bytes(100) + blk)* 20; r = r[:n]
    elif kind == 5:
        a = bytearray(corpus.frame_bytes(int(rng.integers(0, 2000)))[:n])
        for k in range(0, len(a), 97): a[k] = int(rng.integers(0, 256))
        r = bytes(a)
    elif kind == 6:
        parts, tot = [], 0
        while tot < n:
            m = int(rng.integers(1, 3000)); parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0,256))]) * m); tot += m
        r = b"".join(parts)[:n]
    else: r = bytes(rng.integers(0, 5, n, dtype=np.uint8))
    raws.append(r)
frames = [ref.compress(r, level=3, flags=7 if i % 2 else 5) for i, r in enumerate(raws)]
t0 = time.time()
outs, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in raws], n_blocks=3, chunk=0)
bad = sum(1 for r, o, s in zip(raws, outs, st) if s or o != r)
print("decode stress", len(raws), "bad", bad, "fallback", nfb, "%.1fs" % (time.time() - t0))
