"""BASELINE.json configs[1] + configs[2] at FULL size inside the GPU suite (65 536-frame parity lived only in bench.py's own
asserts): 65 536 x 128 KiB Silesia-like frames per direction through the device-resident batch calls, every frame checked -- decode against
the generated input byte for byte, encode against libzstd 1.5.7's frame byte for byte (frames made by native host threads through
bench.compress_on_host, i.e. ZSTD_compressStream2(e_end) like c-ext/compressor.c:1035-1043)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_65536_frames_both_directions_every_frame():
    import torch
    import bench
    from tests import reflib
    from tests.corpus import Corpus
    from zstandard_amd.device import DeviceBatchContext
    assert reflib.have_ref(), "no libzstd 1.5.7 to check against"
    F, item = 65536, 131072
    dev = torch.device("cuda", 0)
    raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
    torch.cuda.synchronize()
    raw_np = raw.cpu().numpy()
    frames, csizes = bench.compress_on_host(raw_np, item)
    assert len(frames) == F
    job = bench.Job(1, dev)
    ctx = DeviceBatchContext()
    # decode: run_decompress asserts status == 0, sizes == 128 KiB and dst == raw for ALL frames
    bench.run_decompress(job, ctx, frames, csizes, raw, item, 1, 0)
    ctx.close()
    torch.cuda.empty_cache()
    # encode: run_compress compares every one of the 65 536 frames with libzstd's
    ctx = DeviceBatchContext()
    _, total, _ = bench.run_compress(job, ctx, raw, frames, item, 1, 0)
    assert total == int(csizes.sum())
    ctx.close()
