"""Emulator stress of the encoder on SOURCES OF SEVERAL BLOCKS against libzstd 1.5.7: python tests/stress_emu_encode_blocks.py SEED [LEVEL].
The flat match kernel searches such a source on an assumption about how its blocks end up (ZeMbBlock, zhip_format.hpp); the inputs here are
made to break it now and then -- incompressible stretches (raw blocks, negative savings), runs of one byte (RLE blocks), data that changes
character inside a block (the pre-splitter), a block that repeats the one before, tails of a few bytes -- every frame must equal libzstd's
(ZSTD_compress_frameChunk, zstd/zstd.c:27545).  Not collected by pytest; bounded version: test_emu_kernels.py."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus

B = 131072


def piece(rng, corpus, n):
    k = int(rng.integers(0, 7))
    if k == 0: return b"".join(corpus.frame_bytes(int(rng.integers(0, 2000)) + j) for j in range(n // B + 1))[:n]
    if k == 1: return rng.bytes(n)
    if k == 2: return bytes([int(rng.integers(0, 256))]) * n
    if k == 3: return bytes(rng.integers(0, 4, n, dtype=np.uint8))
    if k == 4:
        blk = rng.bytes(int(rng.integers(20, 3000))); return (blk * (n // len(blk) + 1))[:n]
    if k == 5:
        a = bytearray(b"".join(corpus.frame_bytes(int(rng.integers(0, 2000)) + j) for j in range(n // B + 1))[:n])
        for q in range(0, len(a), int(rng.integers(50, 2000))): a[q] = int(rng.integers(0, 256))
        return bytes(a)
    return (corpus.frame_bytes(int(rng.integers(0, 2000)))[:int(rng.integers(100, 5000))] + rng.bytes(int(rng.integers(1, 200)))) * (n // 100 + 1)


def make(rng, corpus):
    total = int(rng.choice([B + int(rng.integers(1, 12)), 2 * B + int(rng.integers(0, 9)), int(rng.integers(B + 1, 3 * B)), int(rng.integers(3 * B, 9 * B))]))
    parts, tot = [], 0
    while tot < total:
        n = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, B)), B, int(rng.integers(B, 3 * B))]))
        p = piece(rng, corpus, n)[:n]
        if rng.integers(0, 6) == 0 and parts: p = parts[-1][:n]          # the stretch before, again (matches that span a block)
        parts.append(p); tot += len(p)
    return b"".join(parts)[:total]


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
    rng = np.random.default_rng(seed)
    raws = [make(rng, corpus) for _ in range(12)] + [corpus.frame_bytes(3)[:int(rng.integers(1, B))]]
    flags = int(rng.choice([5, 7, 4]))
    want = [ref.compress(r, level=level, flags=flags) for r in raws]
    t0 = time.time(); r0, s0 = emu.stat(9), emu.stat(8)
    slots = int(rng.choice([1, 1, 1, 3]))                          # 3: so few block slots per frame that the larger sources are not laid out at all
    emu.set_mb_compress(slots)
    outs, st = emu.compress_batch(raws, level=level, flags=flags, n_blocks=2, pipeline=True, chunk=int(rng.choice([0, 5])))
    emu.set_mb_compress(1)
    bad = 0
    for i, (o, w) in enumerate(zip(outs, want)):
        if st[i] or o != w: bad += 1; print("MISMATCH", seed, i, len(raws[i]), st[i], len(o), len(w))
    print("encode-blocks stress", seed, "sources", len(raws), "bad", bad, "searched flat", emu.stat(8) - s0, "redone", emu.stat(9) - r0, "%.1fs" % (time.time() - t0))
    sys.exit(1 if bad else 0)
