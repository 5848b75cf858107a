"""Emulator stress of DICTIONARY compression and decompression against libzstd 1.5.7: python tests/stress_emu_dict.py SEED.
Random dictionaries (trained / raw content, 200 bytes - 110 KiB), random sources on both sides of libzstd's attach cutoffs and of one
block (attach mode, table-copy mode, multi-block frames: DESIGN.md 4.3), levels 3 / 1 / -3, both kernel forms; the frames must be
libzstd's, and must decode -- through the emulated decode pipeline with the same dictionary -- to the sources.
Not collected by pytest; the bounded versions live in test_emu_kernels.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus

emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
pool = corpus.frame_list(40 * (seed % 20), 24)
kind = int(rng.integers(0, 4))
if kind == 0: dd = ref.train_dictionary(int(rng.integers(2000, 112641)), [f[j * 4096:(j + 1) * 4096] for f in pool for j in range(16)])
elif kind == 1: dd = pool[3][1000:1000 + int(rng.integers(200, 112640))]
elif kind == 2: dd = ref.train_dictionary(8192, [f[j * 2048:(j + 1) * 2048] for f in pool[:8] for j in range(24)])
else: dd = (rng.bytes(300) + pool[5][:3000]) * int(rng.integers(1, 12))
raws = []
for i in range(14):
    k = i % 7
    n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 8192), rng.integers(8192, 16385), rng.integers(16385, 131073), rng.integers(131073, 300000)]))
    if k in (0, 1): r = (pool[int(rng.integers(0, 24))] + pool[int(rng.integers(0, 24))] + pool[int(rng.integers(0, 24))])[:n]
    elif k == 2: r = (dd[-min(len(dd), 5000):] + rng.bytes(100) + pool[1])[:n]                     # starts as the dictionary ends
    elif k == 3: r = rng.bytes(n)
    elif k == 4:
        a = bytearray((pool[7] + pool[8] + pool[9])[:n])
        for q in range(0, len(a), 211): a[q] = int(rng.integers(0, 256))
        r = bytes(a)
    elif k == 5: r = (dd[len(dd) // 3: len(dd) // 3 + 700] * (n // 700 + 1))[:n]
    else: r = bytes(rng.integers(0, 4, n, dtype=np.uint8))
    raws.append(r)
bad = 0
t0 = time.time()
for level in (3, 1, -3):
    want = []
    for r in raws:
        try: want.append(ref.compress(r, level=level, dict_data=dd))
        except RuntimeError: want.append(None)
    for pipe in (True, False):
        outs, st = emu.compress_batch(raws, level=level, flags=5, n_blocks=2, pipeline=pipe, dict_data=dd)
        for i, (o, w) in enumerate(zip(outs, want)):
            if w is None: continue
            if st[i] == 0 and o != w: bad += 1; print("MISMATCH", seed, level, pipe, i, len(raws[i]), len(o), len(w))
            elif st[i] != 0: print("refused", seed, level, pipe, i, len(raws[i]), st[i])
    assert emu.set_ddict(dd) == 0
    frames = [w for w in want if w is not None]; srcs = [r for r, w in zip(raws, want) if w is not None]
    outs, st, nfb = emu.decompress_pipeline(frames, [len(r) for r in srcs], n_blocks=2)
    for i, (o, r) in enumerate(zip(outs, srcs)):
        if st[i] or o != r: bad += 1; print("DECODE MISMATCH", seed, level, i, len(r), st[i])
    emu.set_ddict(None)
print("dict stress", seed, "kind", kind, "dict", len(dd), "bad", bad, "%.1fs" % (time.time() - t0))
sys.exit(1 if bad else 0)
