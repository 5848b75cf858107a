"""The flat match kernel with two against four probes per trip, by batch size (analysis aid, test infrastructure; run on the GPU box).
For every size a fresh context per setting (ZHIP_FLAT4_MAX is read when a context is created; ZHIP_E1LDS_MAX=0 keeps the LDS-source kernel out of the
way), three timed calls after a warm-up, the match kernel's average launch time from the library's HIP-event timers; the compressed sizes of both
settings must agree (the frames themselves are checked by the GPU suite).
usage: python tests/tools/flat4_sweep.py [sizes=1024,4096,8192,16384,32768,65536]"""
import importlib
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.corpus import Corpus
import zstandard_amd  # noqa: F401 -- the alias module that makes the hyphenated package importable
dev_mod = importlib.import_module("zstandard_amd.device")

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,4096,8192,16384,32768,65536").split(",")]
Fmax, item = max(sizes), 131072
dev = torch.device("cuda:0")
raw = Corpus(device=dev, mix="silesia").frames(0, Fmax, chunk=256)
bound = (item + (item >> 8) + 64 + 15) & ~15
os.environ["ZHIP_E1LDS_MAX"] = "0"


def segs(offsets, lengths):
    s = np.zeros((len(lengths), 2), dtype=np.int64); s[:, 0] = offsets; s[:, 1] = lengths
    return torch.from_numpy(s).to(dev)


dst = torch.zeros(Fmax * bound, dtype=torch.uint8, device=dev)
src = raw.reshape(-1)
for F in sizes:
    src_segs = segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64))
    dst_segs = segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64))
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    status = torch.zeros(F, dtype=torch.int32, device=dev)
    res = {}
    for name, fmax in (("two probes", "0"), ("four probes", str(F))):
        os.environ["ZHIP_FLAT4_MAX"] = fmax
        ctx = dev_mod.DeviceBatchContext()
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ctx.kernel_time(8); ctx.kernel_time(6)
        for _ in range(3):
            ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time(8); e2, _ = ctx.kernel_time(6)
        assert int(status.abs().max().item()) == 0
        res[name] = (ms, int(out_sizes.sum().item()), e2)
        ctx.close()
    a, b = res["two probes"], res["four probes"]
    print("%6d sources x 128 KiB: match kernel %8.2f ms with two probes per trip, %8.2f with four (%+.1f %%); entropy kernel %.2f ms; compressed bytes %s"
          % (F, a[0], b[0], 100.0 * (b[0] / a[0] - 1.0), a[2], "same" if a[1] == b[1] else "DIFFERENT"), flush=True)
