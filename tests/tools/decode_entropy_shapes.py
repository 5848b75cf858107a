"""Decode rate of batches whose frames are ALL of one unusual shape (robustness check of the compact arena's common budget, round 5; test infrastructure; run on the GPU box):
Huffman-only data (16-symbol noise: ~123 KiB of literals per frame, few sequences), match-only data (4-symbol noise: ~19 K sequences, few literals), 2-symbol noise, and
the bench corpus for reference. 8 192 frames of 128 KiB each, device-resident, every byte compared; printed: GB/s and how many frames went to the generic kernel's list."""
import importlib
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tests.corpus import Corpus
import zstandard_amd  # noqa: F401
dev_mod = importlib.import_module("zstandard_amd.device")

F, item = 8192, 131072
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
shapes = {"corpus": Corpus(device=dev, mix="silesia").frames(0, F, chunk=256),
          "noise16 (Huffman only)": torch.randint(0, 16, (F, item), dtype=torch.uint8, device=dev, generator=g),
          "noise4 (matches only)": torch.randint(0, 4, (F, item), dtype=torch.uint8, device=dev, generator=g),
          "noise2": torch.randint(0, 2, (F, item), dtype=torch.uint8, device=dev, generator=g)}
for name, raw in shapes.items():
    frames, csizes = bench.compress_on_host(raw.cpu().numpy(), item)
    sec = bench.frame_sections(frames, 256)
    offs = np.zeros(F, dtype=np.int64); offs[1:] = np.cumsum(csizes)[:-1]
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    src_segs = bench.segs(offs, csizes, dev)
    dst_segs = bench.segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
    dst = torch.zeros(F * item, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev); status = torch.zeros(F, dtype=torch.int32, device=dev)
    ctx = dev_mod.DeviceBatchContext()
    ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status); torch.cuda.synchronize()
    for k in bench.DEC_KERNELS:
        ctx.kernel_time(k)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    kt = {ctx.kernel_name(k).replace("zhip_decode_", "").replace("_kernel", ""): round(ctx.kernel_time(k)[0], 3) for k in bench.DEC_KERNELS}
    ok = bool(int(status.abs().max().item()) == 0 and torch.equal(dst.view(F, item), raw))
    print("%-24s ratio %5.2f  literals %6.0f B  sequences %6.0f per frame  ->  %7.1f GB/s (%.2f ms)  %s  generic-kernel ms %.3f  %s" % (
        name, F * item / float(csizes.sum()), sec["lit"] + sec["rawlit"], sec["nbseq"], F * item / ms / 1e6, ms, kt, kt.get("frames", 0), "exact" if ok else "WRONG"), flush=True)
    ctx.close()
