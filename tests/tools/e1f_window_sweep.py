"""The flat match kernel with and without the lanes' LDS source windows (ze_dfast_flat_w, round 6), by launch size (analysis aid, test infrastructure;
run on the GPU box). For every size a fresh context per setting (ZHIP_E1F_WIN is read when a context is created; ZHIP_E1LDS_MAX=0 keeps the LDS-source
kernel out of the way, ZHIP_E1F_PICK=0 the placement pick -- both forms then run on the same kind of allocation order), three timed calls after a warm-up,
the match kernel's average launch time from the library's HIP-event timers; every compressed size of both settings must agree (the frames themselves are
checked against libzstd by the GPU suite).   usage: python tests/tools/e1f_window_sweep.py [sizes=8192,32768,65536,131072] [settings=0,1]"""
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

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8192,32768,65536,131072").split(",")]
settings = (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")
Fmax, item = max(sizes), 131072
dev = torch.device("cuda:0")
raw = Corpus(device=dev, mix="silesia").frames(0, Fmax, chunk=256)
bound = (item + (item >> 8) + 64 + 15) & ~15
os.environ["ZHIP_E1LDS_MAX"] = "0"
os.environ.setdefault("ZHIP_E1F_PICK", "0")


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
    for w in settings:
        os.environ["ZHIP_E1F_WIN"] = w
        ctx = dev_mod.DeviceBatchContext()
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ctx.kernel_time(8); ctx.kernel_time(6)
        for _ in range(3):
            ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time(8); e2, _ = ctx.kernel_time(6)
        assert int(status.abs().max().item()) == 0
        res[w] = (ms, out_sizes.clone(), e2)
        ctx.close()
        torch.cuda.empty_cache()
    line = "%6d sources x 128 KiB: match kernel" % F
    for w in settings:
        line += "  %s %8.2f ms" % ("window" if w == "1" else "memory", res[w][0])
    if len(settings) == 2:
        a, b = res[settings[0]], res[settings[1]]
        line += "  (%+.1f %%); compressed sizes %s" % (100.0 * (b[0] / a[0] - 1.0), "same" if bool((a[1] == b[1]).all().item()) else "DIFFERENT")
    print(line + "; entropy kernel %.2f ms" % res[settings[-1]][2], flush=True)
