"""Decode-pipeline kernel times of several builds of the library on ONE set of frames (A/B aid, test infrastructure; run on the GPU box).
The frames of the bench workload are prepared once and kept in /tmp; every library named on the command line (paths, or variant names of
csrc/build_variants.sh; "product" = csrc/libzstd_hip.so) then decodes them `steps` times in its OWN process (ZHIP_LIB: two builds cannot share a
process -- the extension loads the library with global symbols, a second build's kernel stubs would bind to the first's). Printed per library:
the step's wall time, the kernels' average launch times from the library's HIP-event timers, and whether the output equals the input (diagnostic
variants that produce wrong bytes on purpose say "differs").
usage: python tests/tools/decode_variants_ab.py [--frames 65536] [--steps 5] [--rounds 1] lib [lib ...]"""
import argparse
import os
import subprocess
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--child", default=None)
ap.add_argument("libs", nargs="*")
args = ap.parse_args()
F, item = args.frames, 131072
STASH = "/tmp/zhip_ab_%d" % F
import torch
import bench
from tests.corpus import Corpus
dev = torch.device("cuda", 0)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
if args.child is None:
    frames, csizes = bench.compress_on_host(raw.cpu().numpy(), item)
    np.save(STASH + "_blob.npy", np.frombuffer(b"".join(frames), dtype=np.uint8)); np.save(STASH + "_sizes.npy", csizes)
    del raw; torch.cuda.empty_cache()
    csrc = os.path.join(ROOT, "python-zstandard_amd", "csrc")
    for rnd in range(args.rounds):
        for name in args.libs:
            path = name if os.path.sep in name else os.path.join(csrc, "libzstd_hip.so" if name == "product" else "libzstd_hip_%s.so" % name)
            env = dict(os.environ); env["ZHIP_LIB"] = path
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--frames", str(F), "--steps", str(args.steps), "--child", name], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("step")]
            print("round %d  %-12s %s" % (rnd, name, line[0] if line else "FAILED: " + p.stderr[-300:]), flush=True)
    sys.exit(0)
import importlib
import zstandard_amd  # noqa: F401
dev_mod = importlib.import_module("zstandard_amd.device")
blob, csizes = np.load(STASH + "_blob.npy"), np.load(STASH + "_sizes.npy")
offs = np.zeros(F, dtype=np.int64); offs[1:] = np.cumsum(csizes)[:-1]
hold_gib = int(os.environ.get("ZHIP_AB_HOLD_GIB", "0"))       # placement aid: a dummy allocation held in front of the run's own, so that everything after it lands on other physical pages
hold = torch.empty(hold_gib << 30, dtype=torch.uint8, device=dev) if hold_gib else None
src = torch.from_numpy(blob).to(dev)
src_segs = bench.segs(offs, csizes, dev)
dst_segs = bench.segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
dst = torch.zeros(F * item, dtype=torch.uint8, device=dev)
out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
status = torch.zeros(F, dtype=torch.int32, device=dev)
free0, _ = torch.cuda.mem_get_info()
ctx = dev_mod.DeviceBatchContext()
ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
for k in bench.DEC_KERNELS:
    ctx.kernel_time(k)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
kt = {ctx.kernel_name(k).replace("zhip_decode_", "").replace("_kernel", ""): round(ctx.kernel_time(k)[0], 3) for k in bench.DEC_KERNELS}
same = bool(int(status.abs().max().item()) == 0 and torch.equal(dst.view(F, item), raw))
print("step %7.3f ms  %s  scratch %.2f GiB  output %s" % (ms, kt, (free0 - free1) / 2**30, "equals the input" if same else "differs (diagnostic build?)"), flush=True)
ctx.close()
