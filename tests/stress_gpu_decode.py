"""GPU stress of the decode pipeline on LARGE MIXED batches against libzstd 1.5.7: python tests/stress_gpu_decode.py SEED [FRAMES].

The emulator's damaged-frame fuzz (tests/stress_emu_corrupt.py) runs a few dozen frames per launch; on the GPU the kernels that only start with
large chunks -- K0, the lane-per-frame parsers (from 6 144 frames per chunk), KX, the lane-per-frame checksum pass -- and the side stream meet
whole, damaged, checksummed and short frames in ONE launch here: FRAMES (default 8 192) frames made of ~300 distinct libzstd frames (levels
-5 ... 19, every third with a content checksum, sources of 1 byte ... 128 KiB of the emulator tests' kinds), three of four damaged the way the
emulator's fuzz damages them, every fourth whole, one in sixteen with a slot that is too small. What must hold (zstd/zstd.c:44174
ZSTD_decompressFrame is the reference's behaviour on the same bytes):
  * a whole frame decodes to its source whatever its neighbours are;
  * a frame we accept decodes to exactly what libzstd decodes it to, and libzstd accepts it;
  * a frame libzstd rejects is rejected (the documented exception the other way round -- DESIGN.md section 2 -- is counted as "stricter").
Two launches per seed: through a context as it comes (K0 by the chunk's size) and through one with ZHIP_K0_MIN=1000000 (K1's own parsers):
status and bytes must agree frame by frame. Not collected by pytest."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import reflib
from tests.corpus import Corpus
from tests.stress_emu_corrupt import damage, make_raws


def main():
    import importlib
    import torch
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    dev_mod = importlib.import_module("zstandard_amd.device")
    ref = reflib.checker(); corpus = Corpus()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    raws = make_raws(rng, corpus, 240, small=False) + make_raws(rng, corpus, 60, small=True) + [b"", b"a", bytes(131072), rng.bytes(131072), rng.bytes(7)]
    levels = [3, 3, 3, 1, 2, 5, -1, -5, 4, 7, 3, 9, 3, 12, 3, 19]
    base = []
    for i, r in enumerate(raws):
        lv = levels[i % len(levels)]
        if lv >= 12 and len(r) > 40000: r = raws[i] = r[:40000]                    # (libzstd's high levels take their time; the frames' shapes are what matters)
        base.append(ref.compress(r, level=lv, flags=7 if i % 3 == 0 else 5))
    frames, sizes, whole = [], [], []
    for k in range(count):
        i = int(rng.integers(0, len(base)))
        f, n = base[i], len(raws[i])
        if k % 4 != 3 and len(f) > 12:
            f = damage(rng, f)
        elif k % 16 == 7 and n > 1:
            n = int(rng.integers(0, n))                                             # a whole frame whose slot is too small
        frames.append(f); sizes.append(n); whole.append(i if (f is base[i] and n == len(raws[i])) else -1)
    n = count
    dev = torch.device("cuda", 0)
    csz = np.array([len(f) for f in frames], dtype=np.int64)
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    ss = np.zeros((n, 2), dtype=np.int64); ss[1:, 0] = np.cumsum(csz)[:-1]; ss[:, 1] = csz
    cap = np.array([(s + 64 + 15) // 16 * 16 for s in sizes], dtype=np.int64)
    ds = np.zeros((n, 2), dtype=np.int64); ds[1:, 0] = np.cumsum(cap)[:-1]; ds[:, 1] = np.array(sizes, dtype=np.int64)
    res = {}
    for form in ("as it comes", "K1's own parsers"):
        if form != "as it comes":
            os.environ["ZHIP_K0_MIN"] = "1000000"
        try:
            ctx = dev_mod.DeviceBatchContext()
        finally:
            os.environ.pop("ZHIP_K0_MIN", None)
        dst = torch.full((int(cap.sum()),), 0xA5, dtype=torch.uint8, device=dev)
        out_sizes = torch.zeros(n, dtype=torch.int64, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
        for _ in range(2):                                                          # twice: the second launch runs on the first one's scratch
            ctx.decompress(src, torch.from_numpy(ss).to(dev), dst, torch.from_numpy(ds).to(dev), out_sizes, status)
            torch.cuda.synchronize()
        res[form] = (status.cpu().numpy().copy(), out_sizes.cpu().numpy().copy(), dst.cpu().numpy().copy())
        ctx.close()
    st, osz, out = res["as it comes"]; st1, osz1, out1 = res["K1's own parsers"]
    tot = dict(frames=n, whole=0, accepted=0, rejected=0, stricter=0, wrong=0, missed=0, neighbours_bad=0, forms_differ=0, slot_overrun=0)
    seen = {}; where = {}
    def note(k, i):
        tot[k] += 1; where.setdefault(k, []).append((i, int(st[i]), int(st1[i]), len(frames[i]), sizes[i]))
    for i in range(n):
        o = int(ds[i, 0]); ok = st[i] == 0
        got = out[o:o + int(osz[i])].tobytes() if ok else None
        if (st[i] == 0) != (st1[i] == 0) or (ok and (osz[i] != osz1[i] or got != out1[o:o + int(osz1[i])].tobytes())): note("forms_differ", i)
        if not (out[o + sizes[i]: o + int(cap[i])] == 0xA5).all(): note("slot_overrun", i)          # nothing is written behind a frame's slot
        if whole[i] >= 0:
            tot["whole"] += 1
            if not ok or got != raws[whole[i]]: note("neighbours_bad", i)
            continue
        key = (frames[i], sizes[i])
        if key not in seen:
            try:
                want = ref.decompress(frames[i], sizes[i])
                if len(want) != sizes[i]: want = None
            except RuntimeError:
                want = None
            seen[key] = want
        want = seen[key]
        if ok:
            tot["accepted"] += 1
            if want is None: note("missed", i)
            elif got != want: note("wrong", i)
        else:
            tot["rejected"] += 1
            if want is not None: tot["stricter"] += 1
    print("gpu decode stress seed", seed, tot, "%.1fs" % (time.time() - t0), {k: v[:6] for k, v in where.items()} or "")
    return 1 if tot["wrong"] or tot["missed"] or tot["neighbours_bad"] or tot["forms_differ"] or tot["slot_overrun"] else 0


if __name__ == "__main__":
    sys.exit(main())
