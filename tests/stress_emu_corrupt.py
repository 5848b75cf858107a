"""Emulator fuzz of the decode pipeline on DAMAGED frames: python tests/stress_emu_corrupt.py SEED [ROUNDS] [dict].

Every frame is a libzstd 1.5.7 frame with a few bits flipped, bytes overwritten, a piece cut out or its tail dropped.  What must hold
(the reference's behaviour on the same bytes is libzstd's, zstd/zstd.c:44174 ZSTD_decompressFrame):
  * no out-of-bounds access -- run it under the AddressSanitizer build (tests/emu/build_asan.sh) to see them;
  * a frame we accept decodes to exactly what libzstd decodes it to;
  * a frame libzstd rejects is rejected (the one documented exception the other way round, DESIGN.md section 2: damaged Huffman streams
    that libzstd's fast loop lets through are refused here -- counted as "stricter");
  * undamaged neighbours in the same batch decode -- also next to a whole frame whose output slot is too small for it.
Not collected by pytest; the bounded version is tests/test_emu_kernels.py::test_damaged_frames_through_the_emulated_pipeline."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus


def damage(rng, frame):
    b = bytearray(frame)
    kind = int(rng.integers(0, 6))
    if kind == 0:                                   # a few bit flips anywhere after the magic
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:                                 # bit flips in the headers (frame header, block header, literals / sequences headers)
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(4, min(len(b), 24)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2:                                 # tail dropped
        del b[int(rng.integers(5, len(b))):]
    elif kind == 3:                                 # a run of bytes overwritten
        at = int(rng.integers(4, len(b))); n = int(rng.integers(1, 40))
        b[at:at + n] = rng.bytes(min(n, len(b) - at))
    elif kind == 4:                                 # a piece cut out of the middle
        at = int(rng.integers(5, len(b))); n = int(rng.integers(1, 64))
        del b[at:at + n]
    else:                                           # the last bytes (end of the sequences bitstream / checksum)
        for _ in range(int(rng.integers(1, 3))):
            b[len(b) - 1 - int(rng.integers(0, min(12, len(b) - 4)))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def make_raws(rng, corpus, count, small, big=False):
    raws = []
    for i in range(count):
        kind = i % 6
        n = int(rng.integers(200, 6000 if small else 131073))
        if big and i % 3 == 0: n = int(rng.integers(131073, 400000))                 # multi-block frames: the generic kernel's
        if kind in (0, 1): r = b"".join(corpus.frame_bytes(int(rng.integers(0, 2000)) + k) for k in range(n // 131072 + 1))[:n]
        elif kind == 2:
            blk = rng.bytes(300); r = ((blk + rng.bytes(800) + blk * 3 + rng.bytes(50)) * (n // 1900 + 1))[:n]
        elif kind == 3:
            a = bytearray(corpus.frame_bytes(int(rng.integers(0, 2000)))[:n])
            for k in range(0, len(a), 61): a[k] = int(rng.integers(0, 256))
            r = bytes(a)
        elif kind == 4: r = bytes(rng.integers(0, 6, n, dtype=np.uint8))
        else:
            parts, tot = [], 0
            while tot < n:
                m = int(rng.integers(1, 900)); parts.append(rng.bytes(m) if rng.integers(0, 2) else bytes([int(rng.integers(0, 256))]) * m); tot += m
            r = b"".join(parts)[:n]
        raws.append(r)
    return raws


def one_round(emu, ref, rng, corpus, count=24, small=False, dict_data=None, n_blocks=3, big=False):
    raws = make_raws(rng, corpus, count, small, big)
    frames = [ref.compress(r, level=int(rng.choice([1, 3, 3, 3, 5])), flags=7 if i % 2 else 5, dict_data=dict_data) for i, r in enumerate(raws)]
    bad = [f if i % 4 == 3 else damage(rng, f) for i, f in enumerate(frames)]          # every fourth frame stays whole
    sizes = [len(r) for r in raws]
    for i in range(2, count, 12):                                                     # a whole frame whose slot is too small: refused, the slot after it untouched
        bad[i] = frames[i]; sizes[i] = int(rng.integers(0, len(raws[i])))
    outs, st, nfb = emu.decompress_pipeline(bad, sizes, n_blocks=n_blocks, chunk=0)
    res = dict(frames=count, accepted=0, rejected=0, stricter=0, wrong=0, missed=0, neighbours_bad=0)
    for i, (f, r) in enumerate(zip(bad, raws)):
        try:
            want = ref.decompress(f, sizes[i], dict_data=dict_data)
            if len(want) != sizes[i]: want = None                                    # the batch call passes the expected size: a mismatch is an error there
        except RuntimeError:
            want = None
        ok = st[i] == 0
        if i % 4 == 3 and (not ok or outs[i] != r): res["neighbours_bad"] += 1
        if ok:
            res["accepted"] += 1
            if want is None: res["missed"] += 1                                      # we accepted what libzstd rejects
            elif outs[i] != want: res["wrong"] += 1
        else:
            res["rejected"] += 1
            if want is not None: res["stricter"] += 1
    return res


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    use_dict = len(sys.argv) > 3 and sys.argv[3] == "dict"
    emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
    rng = np.random.default_rng(seed)
    dict_data = None
    blocks = int(os.environ.get("ZHIP_EMU_BLOCKS", "0"))          # several-block mode of the pipeline: item slots per frame (then every other round has big frames)
    emu.set_blocks(blocks)
    if use_dict:
        samples = [f[:4096] for f in corpus.frame_list(0, 400)]
        dict_data = ref.train_dictionary(16384, samples)
        assert emu.set_ddict(dict_data) == 0
    tot = {}
    t0 = time.time()
    for k in range(rounds):
        r = one_round(emu, ref, rng, corpus, small=use_dict or k % 2 == 1, dict_data=dict_data, big=(k % 4 == 2 or (blocks and k % 2 == 0)))
        for a, b in r.items(): tot[a] = tot.get(a, 0) + b
    print("corrupt stress seed", seed, tot, "%.1fs" % (time.time() - t0))
    sys.exit(1 if tot["wrong"] or tot["missed"] or tot["neighbours_bad"] else 0)
