"""Can the match kernel's fast placement regime be PICKED (analysis aid, test infrastructure; run on the GPU box)?
E1f's time for the same launch differs by up to 20 % with where the driver puts the context's tables (a property of one
allocation, reproducible within a process, not controllable through the allocator). The one lever left: hold several candidate contexts at once
(their allocations cannot be the same memory), time the real kernel on each, keep the best. Per trial: context A; then B and C created while the
earlier ones are still alive (and with a perturbing allocation of a few GiB in between); every context's match-kernel time over two launches of
65 536 x 128 KiB. Reported: the spread inside a trial, and how often the first context was within 3 % of the trial's best.
usage: python tests/tools/e1f_pick_best.py [trials=4]"""
import importlib
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.corpus import Corpus
import zstandard_amd  # noqa: F401
dev_mod = importlib.import_module("zstandard_amd.device")

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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


def measure(ctx):
    ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    torch.cuda.synchronize()
    ctx.kernel_time(8)
    for _ in range(2):
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    torch.cuda.synchronize()
    assert int(status.abs().max().item()) == 0
    return ctx.kernel_time(8)[0]


first_ok = 0
rng = np.random.default_rng(1)
for t in range(trials):
    ctxs, times, pads = [], [], []
    for k in range(3):
        if k:
            pads.append(torch.empty(int(rng.integers(1, 9)) << 30, dtype=torch.uint8, device=dev))
        c = dev_mod.DeviceBatchContext()
        ctxs.append(c); times.append(measure(c))
    again = measure(ctxs[int(np.argmin(times))])                     # is the best one still the best with the others alive?
    best = min(times)
    first_ok += times[0] <= 1.03 * best
    print("trial %d  match kernel ms: A %.1f  B %.1f  C %.1f   best again %.1f   spread %.1f %%" % (t, times[0], times[1], times[2], again, 100 * (max(times) / best - 1)), flush=True)
    for c in ctxs:
        c.close()
    del pads, ctxs
    torch.cuda.empty_cache()
print("first context within 3 %% of the trial's best: %d of %d trials" % (first_ok, trials))
