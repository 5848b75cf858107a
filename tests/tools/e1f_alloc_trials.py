"""E1f's regimes against how its hash tables were allocated, many trials in ONE process (analysis aid, test infrastructure; run on the GPU box).
Every trial creates a fresh device context (the tables are allocated when the first batch is launched), compresses 65 536 x 128 KiB twice and
reads the flat match kernel's average launch time from the library's HIP-event timers. ZHIP_TABLES_VMM (MiB per physical chunk, 0 = one
hipMalloc) is read when a context is created, so it can change from trial to trial.
usage: python tests/tools/e1f_alloc_trials.py [rounds=3] [settings=0,2,64,...]"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
settings = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "2", "64"]
F, item = 65536, 131072
dev = torch.device("cuda:0")
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
bound = (item + (item >> 8) + 64 + 15) & ~15


def segs(offsets, lengths):
    s = np.zeros((len(lengths), 2), dtype=np.int64); s[:, 0] = offsets; s[:, 1] = lengths
    return torch.from_numpy(s).to(dev)


src_segs = segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64))
dst_segs = segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64))
dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
status = torch.zeros(F, dtype=torch.int32, device=dev)
src = raw.reshape(-1)
first = None
for r in range(rounds):
    for sname in settings:
        for k in ("ZHIP_TABLES_VMM", "ZHIP_FLAT3", "ZHIP_FLAT4_MAX"):
            os.environ.pop(k, None)
        if "=" in sname:                                              # NAME=VALUE[+NAME=VALUE]: any knob that is read when a context is created
            for kv in sname.split("+"):
                os.environ[kv.split("=")[0]] = kv.split("=")[1]
        else:
            os.environ["ZHIP_TABLES_VMM"] = sname
        ctx = dev_mod.DeviceBatchContext()
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)        # warm-up: allocations, first touch
        torch.cuda.synchronize()
        ctx.kernel_time(8)
        for _ in range(2):
            ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time(8)
        total = int(out_sizes.sum().item())
        if first is None:
            first = total
        free, tot = torch.cuda.mem_get_info()
        print("round %d  %-22s  E1f %7.2f ms (%d launches)  compressed bytes %s  free VRAM %.1f GiB" % (r, sname if "=" in sname else "ZHIP_TABLES_VMM=" + sname, ms, n, "same" if total == first else "DIFFERENT", free / 2**30), flush=True)
        assert int(status.abs().max().item()) == 0
        ctx.close()
