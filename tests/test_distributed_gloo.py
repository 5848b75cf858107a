"""world_size-2 CPU test (gloo) of the N>1 path: reference partition rule + segment-table all-gather.
(The data path itself needs no collective: frames are independent and results stay sharded.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    par = importlib.import_module("python-zstandard_amd.parallel")
    sizes = [1000 + 37 * i for i in range(101)]                      # every rank sees the same item list
    lo, hi = par.my_shard(sizes)
    local = torch.tensor([s // 3 + 1 for s in sizes[lo:hi]], dtype=torch.int64)   # stand-in for per-frame output sizes
    per_rank = par.gather_segment_table(local)
    segs = par.global_segments(per_rank)
    q.put((rank, lo, hi, [t.tolist() for t in per_rank], segs[:3], len(segs)))
    dist.destroy_process_group()


def test_sharding_and_segment_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, lo0, hi0, tab0, head0, n0), (r1, lo1, hi1, tab1, head1, n1) = res
    assert lo0 == 0 and hi0 == lo1 and hi1 == 101                      # contiguous cover
    sizes = [1000 + 37 * i for i in range(101)]
    assert abs(sum(sizes[lo0:hi0]) - sum(sizes[lo1:hi1])) <= max(sizes) * 2   # balanced by bytes
    assert tab0 == tab1 and n0 == n1 == 101                           # both ranks hold the same global table
    assert tab0[0] == [s // 3 + 1 for s in sizes[lo0:hi0]] and tab0[1] == [s // 3 + 1 for s in sizes[lo1:hi1]]
    assert head0[0] == (0, 0, sizes[0] // 3 + 1)
