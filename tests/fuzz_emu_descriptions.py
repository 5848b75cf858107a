"""The kernels' twin of tests/fuzz_oracle_descriptions.py: bit flips confined to the table-description regions of real frames, 200 frames per
launch through the emulated decode pipeline, against libzstd 1.5.7. python tests/fuzz_emu_descriptions.py SEED [CASES]; MISSED / WRONG must stay
at zero ('stricter' is DESIGN.md section 2's Huffman-stream exception). Not collected by pytest."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from tests import reflib, emulib
from tests.corpus import Corpus
ref = reflib.RefZstd(); emu = emulib.Emu(); corpus = Corpus()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(seed)
srcs = []
for i in range(24):
    n = int(rng.integers(600, 9000)); k = i % 4
    base = corpus.frame_bytes(int(rng.integers(0, 3000)))
    if k == 0: r = base[:n]
    elif k == 1: r = bytes(rng.integers(0, 12, n, dtype=np.uint8))
    elif k == 2: r = (base[:300] + bytes(rng.integers(97, 105, 80, dtype=np.uint8))) * (n // 380 + 1)
    else: r = bytes((np.frombuffer(base[:n], dtype=np.uint8) & 0x3F).tobytes())
    r = r[:n]
    f = ref.compress(r, level=int(rng.choice([1, 3, 3, 5, 9, 19])))
    fhd = f[4]; single = (fhd >> 5) & 1; fcs = fhd >> 6
    pos = 5 + (0 if single else 1) + [0, 1, 2, 4][fhd & 3] + ([1 if single else 0, 2, 4, 8][fcs])
    bh = int.from_bytes(f[pos:pos + 3], "little"); btype = (bh >> 1) & 3; bs = bh >> 3
    if btype != 2: continue
    b0 = pos + 3; lt = f[b0] & 3; fmt = (f[b0] >> 2) & 3
    if lt < 2:
        lh = 1 if fmt in (0, 2) else 2 if fmt == 1 else 3
        regen = f[b0] >> 3 if lh == 1 else (int.from_bytes(f[b0:b0 + 2], "little") >> 4) if lh == 2 else (int.from_bytes(f[b0:b0 + 3], "little") >> 4)
        lit_end = b0 + lh + (regen if lt == 0 else 1); huf = None
    else:
        v = int.from_bytes(f[b0:b0 + 5], "little")
        if fmt < 2: lh, csize = 3, (v >> 14) & 0x3FF
        elif fmt == 2: lh, csize = 4, (v >> 18) & 0x3FFF
        else: lh, csize = 5, (v >> 22) & 0x3FFFF
        lit_end = b0 + lh + csize; huf = (b0, min(b0 + lh + 140, lit_end))
    srcs.append((f, len(r), huf, (lit_end, min(lit_end + 90, b0 + bs))))
cats = {}; t0 = time.time(); done = 0
while done < N:
    batch = []
    for _ in range(200):
        f, n, huf, seq = srcs[int(rng.integers(0, len(srcs)))]
        lo, hi = huf if (huf and rng.integers(0, 2)) else seq
        b = bytearray(f)
        for _ in range(int(rng.integers(1, 3))): b[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
        batch.append((bytes(b), n))
    outs, st, nfb = emu.decompress_pipeline([x[0] for x in batch], [x[1] for x in batch], n_blocks=3, chunk=0)
    for (b, n), o, s in zip(batch, outs, st):
        try: w = ref.decompress(b, n)
        except RuntimeError: w = None
        if w is not None and len(w) != n: w = None
        got = o if s == 0 else None
        if got == w: continue
        cat = "stricter" if (w is not None and got is None) else "MISSED" if w is None else "WRONG"
        cats[cat] = cats.get(cat, 0) + 1
        if cat != "stricter" and cats[cat] <= 5: print(cat, s, b.hex(), n)
    done += 200
print("emu description fuzz seed", seed, "cases", done, cats, "%.1fs" % (time.time() - t0))

sys.exit(1 if cats.get("MISSED", 0) + cats.get("WRONG", 0) else 0)
