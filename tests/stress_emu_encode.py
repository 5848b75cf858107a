"""Emulator stress of the encoder (product kernels under tests/emu) against libzstd 1.5.7: python tests/stress_emu_encode.py SEED [p|n] [LEVEL]
(p = two-kernel form with the flat match kernel, l = the same with the LDS-source match kernel of small batches, n = fused kernel). Not collected by pytest; the bounded versions live in test_emu_kernels.py."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus
emu = emulib.Emu()
ref = reflib.RefZstd()
corpus = Corpus()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
pipeline = (sys.argv[2] in ('p', 'l')) if len(sys.argv) > 2 else True
if len(sys.argv) > 2 and sys.argv[2] == 'l': emu.lib.emu_set_e1lds_max(1000)
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
raws = []
for i in range(40):
    kind = i % 8
    n = int(rng.integers(1, 131073)) if i % 3 else int(rng.integers(1, 3000))
    if kind == 0: r = corpus.frame_bytes(int(rng.integers(0, 1000)))[:n]
    elif kind == 1: r = rng.bytes(n)
    elif kind == 2: r = bytes(rng.integers(0, 3, n, dtype=np.uint8))
    elif kind == 3: r = (b"abcdefgh" * (n // 8 + 1))[:n]
    elif kind == 4:
        base = rng.bytes(500); r = (base * (n // 500 + 1))[:n]
    elif kind == 5:
        a = bytearray(corpus.frame_bytes(int(rng.integers(0, 1000)))[:n])
        for k in range(0, len(a), 997): a[k] = int(rng.integers(0, 256))
        r = bytes(a)
    elif kind == 6: r = b"\0" * n
    else:
        parts = []
        tot = 0
        while tot < n:
            m = int(rng.integers(1, 5000))
            parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0,256))]) * m); tot += m
        r = b"".join(parts)[:n]
    raws.append(r)
raws += [corpus.frame_bytes(5), corpus.frame_bytes(6)]
t0 = time.time()
outs, st = emu.compress_batch(raws, level=level, flags=5, n_blocks=2, pipeline=pipeline)
bad = 0
for i, (r, o) in enumerate(zip(raws, outs)):
    e = ref.compress(r, level=level)
    if st[i] or o != e:
        bad += 1
        print("MISMATCH", i, len(r), st[i], len(o), len(e))
emu.lib.emu_stat.restype = __import__("ctypes").c_long
print("done", len(raws), "bad", bad, "%.1fs" % (time.time() - t0), "flat frames", emu.lib.emu_stat(15))
