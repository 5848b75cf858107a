"""The match kernel's rate against the frames in flight (analysis aid, test infrastructure; run on the GPU box).
For every launch size the same corpus prefix is compressed by ONE launch of the flat match kernel (ZHIP_ECHUNK_MAX = the size), twice after a
warm-up call; printed: the match kernel's and the entropy kernel's average launch times from the library's HIP-event timers, the per-frame cost,
and whether the first 65 536 frames came out byte-identical to the 65 536-frame launch's (sizes + a 64-bit sum of every frame's bytes); 64 evenly
spaced frames of the largest launch are compared with libzstd 1.5.7's.
usage: python tests/tools/e1f_frames_in_flight.py [sizes=65536,131072,196608,262144]"""
import importlib
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.corpus import Corpus
from tests import reflib
import zstandard_amd  # noqa: F401 -- the alias module that makes the hyphenated package importable
dev_mod = importlib.import_module("zstandard_amd.device")

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,131072,196608,262144").split(",")]
item = 131072
dev = torch.device("cuda:0")
bound = (item + (item >> 8) + 64 + 15) & ~15
free, tot = torch.cuda.mem_get_info()
print("device memory: %.1f GiB free of %.1f GiB" % (free / 2**30, tot / 2**30), flush=True)


def segs(offsets, lengths):
    s = np.zeros((len(lengths), 2), dtype=np.int64); s[:, 0] = offsets; s[:, 1] = lengths
    return torch.from_numpy(s).to(dev)


def frame_sums(dst, out_sizes, F):
    """a 64-bit sum per frame over the bytes it produced (slots are zero beyond: the destination is zeroed before every configuration)"""
    v = dst.view(F, bound)
    out = torch.empty(F, dtype=torch.int64, device=dev)
    for a in range(0, F, 2048):
        b = min(F, a + 2048)
        out[a:b] = v[a:b].to(torch.int64).mul_(torch.arange(1, bound + 1, device=dev, dtype=torch.int64) % 251 + 1).sum(dim=1)
    return out


Fmax = max(sizes)
raw = Corpus(device=dev, mix="silesia").frames(0, Fmax, chunk=256)
torch.cuda.synchronize()
ref_sizes = ref_sums = None
ctx = dst = None
for F in sizes:
    os.environ["ZHIP_ECHUNK_MAX"] = str(max(F, 65536))
    try:
        src = raw[:F].reshape(-1)
        src_segs = segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64))
        dst_segs = segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64))
        dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
        out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
        status = torch.zeros(F, dtype=torch.int32, device=dev)
        ctx = dev_mod.DeviceBatchContext()
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ctx.kernel_time(8); ctx.kernel_time(6)
        import time
        t0 = time.perf_counter()
        for _ in range(2):
            ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 2
        ms, n = ctx.kernel_time(8)
        ms2, n2 = ctx.kernel_time(6)
        assert int(status.abs().max().item()) == 0
        sums = frame_sums(dst, out_sizes, F)
        if ref_sizes is None:
            ref_sizes, ref_sums = out_sizes[:65536].clone(), sums[:65536].clone()
        k = min(F, ref_sizes.numel())
        same = bool(torch.equal(out_sizes[:k], ref_sizes[:k]) and torch.equal(sums[:k], ref_sums[:k]))
        free, _ = torch.cuda.mem_get_info()
        print("F %7d  match kernel %8.2f ms x %d launches per call (%.3f us / frame)  entropy %7.2f ms x %d  call %8.1f ms = %6.2f GB/s  first %d frames %s  free VRAM %.1f GiB"
              % (F, ms, n // 2, ms * (n // 2) / F * 1e3, ms2, n2 // 2, wall * 1e3, F * item / wall / 1e9, k, "identical to the first configuration's" if same else "DIFFERENT", free / 2**30), flush=True)
        if F == Fmax and reflib.have_ref():
            enc = reflib.RefZstd()
            idx = np.linspace(0, F - 1, 64).astype(np.int64)
            hs = out_sizes.cpu().numpy()
            bad = 0
            for i in idx:
                got = bytes(dst.view(F, bound)[int(i), : int(hs[i])].cpu().numpy())
                bad += got != enc.compress(bytes(raw[int(i)].cpu().numpy()))
            print("F %7d  64 evenly spaced frames against libzstd 1.5.7: %d differ" % (F, bad), flush=True)
        ctx.close()
        del dst, ctx
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 -- an allocation that does not fit ends this size, not the run
        print("F %7d  failed: %s: %s" % (F, type(e).__name__, e), flush=True)
        try:
            ctx.close()
        except Exception:  # noqa: BLE001
            pass
        ctx = dst = None
        torch.cuda.empty_cache()
