"""Emulator stress of the DECODERS on whatever libzstd 1.5.7 can produce: python tests/stress_emu_decode_any.py SEED.
Frames of random levels (-5 ... 19), explicit parameters (small windows: headers with a window descriptor, offsets across blocks), with
and without checksum / content size, sources from a few bytes to several blocks -- through the emulated pipeline (single-block frames; again in its several-block mode) and
the generic kernel (everything else). Every frame must decode to its source. Not collected by pytest."""
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
pool = corpus.frame_list(16 * (seed % 40), 8)
raws, frames = [], []
for i in range(20):
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 131073), rng.integers(131073, 420000)]))
    k = int(rng.integers(0, 5))
    if k == 0: r = (pool[int(rng.integers(0, 8))] + pool[int(rng.integers(0, 8))] + pool[int(rng.integers(0, 8))] + pool[int(rng.integers(0, 8))])[:n]
    elif k == 1: r = rng.bytes(n)
    elif k == 2:
        blk = rng.bytes(int(rng.integers(20, 900))); r = ((blk + rng.bytes(int(rng.integers(1, 3000)))) * (n // 20 + 1))[:n]
    elif k == 3: r = bytes(rng.integers(0, 5, n, dtype=np.uint8))
    else:
        parts, tot = [], 0
        while tot < n:
            m = int(rng.integers(1, 40000)); parts.append(rng.bytes(m) if rng.integers(0, 3) == 0 else bytes([int(rng.integers(0, 256))]) * m if rng.integers(0, 2) else pool[int(rng.integers(0, 8))][:m]); tot += m
        r = b"".join(parts)[:n]
    kw = {}
    if rng.integers(0, 3) == 0: kw["window_log"] = int(rng.integers(10, 21))
    if rng.integers(0, 4) == 0: kw["min_match"] = int(rng.integers(3, 8))
    if rng.integers(0, 4) == 0: kw["strategy"] = int(rng.integers(1, 10))
    level = int(rng.choice([-5, -1, 1, 2, 3, 3, 4, 5, 7, 9, 12, 16, 19]))
    flags = int(rng.choice([5, 7, 4, 6, 1]))
    raws.append(r); frames.append(ref.compress_advanced(r, level=level, flags=flags, **kw))
t0 = time.time()
sizes = [len(r) for r in raws]
outs, st, nfb = emu.decompress_pipeline(frames, sizes, n_blocks=3, chunk=0)
bad = sum(1 for r, o, s in zip(raws, outs, st) if s or o != r)
outs2, st2 = emu.decompress_batch(frames, sizes, n_blocks=2)
bad2 = sum(1 for r, o, s in zip(raws, outs2, st2) if s or o != r)
# the several-block mode of the pipeline (the item is a block): enough slots for every frame, then so few that some frames overflow into the generic kernel
for per_frame in (int(rng.integers(5, 12)), 2):
    emu.set_blocks(per_frame)
    outs3, st3, nfb3 = emu.decompress_pipeline(frames, sizes, n_blocks=3, chunk=int(rng.choice([0, 7])))
    emu.set_blocks(0)
    for i, (r, o, s) in enumerate(zip(raws, outs3, st3)):
        if s or o != r: print("BLOCK-MODE MISMATCH", seed, per_frame, i, len(r), s); bad2 += 1
    print("  block mode, %d slots per frame: fallback %d" % (per_frame, nfb3))
for i, (r, o, s) in enumerate(zip(raws, outs, st)):
    if s or o != r: print("PIPELINE MISMATCH", seed, i, len(r), s)
for i, (r, o, s) in enumerate(zip(raws, outs2, st2)):
    if s or o != r: print("GENERIC MISMATCH", seed, i, len(r), s)
print("decode-any stress", seed, "frames", len(raws), "bad", bad + bad2, "fallback", nfb, "%.1fs" % (time.time() - t0))
sys.exit(1 if bad + bad2 else 0)
