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


# ------------------------------------------------------------------------------------------------------------------------------
# the callable sharded path (sharded.py) end to end on CPU: partition -> local batch -> size exchange -> payload all-gatherv ->
# reassembly. The GPU context is replaced by a stand-in that runs the CHECKERS on CPU tensors (test infrastructure: the product's
# default ctx_factory is DeviceBatchContext and needs a GPU).
class _CpuCtx:
    device = torch.device("cpu")

    def __init__(self, dict_data=None, level=3, **kw):
        from tests import reflib
        self.z = reflib.RefZstd() if reflib.have_ref() else reflib.Oracle()
        self.level = level

    def compress(self, src, src_segs, dst, dst_segs, out_sizes, status):
        for i in range(src_segs.shape[0]):
            o, n = (int(v) for v in src_segs[i])
            f = self.z.compress(bytes(src[o:o + n].numpy()), level=self.level)
            d = int(dst_segs[i, 0])
            assert len(f) <= int(dst_segs[i, 1])
            dst[d:d + len(f)] = torch.frombuffer(bytearray(f), dtype=torch.uint8)
            out_sizes[i] = len(f)

    def compact(self, arena, segs, out_sizes, status, offs, dense):
        """what zhip_compact_device does on the GPU (include/zstd_hip.h): the valid prefix of every slot -> its place in the dense arena"""
        for i in range(segs.shape[0]):
            if int(status[i]) == 0:
                o, n, d = int(segs[i, 0]), int(out_sizes[i]), int(offs[i])
                dense[d:d + n] = arena[o:o + n]

    def decompress(self, src, src_segs, dst, dst_segs, out_sizes, status):
        for i in range(src_segs.shape[0]):
            o, n = (int(v) for v in src_segs[i])
            cap = int(dst_segs[i, 1])
            try:
                r = self.z.decompress(bytes(src[o:o + n].numpy()), cap)
            except Exception:
                status[i] = 20                                           # corruption_detected
                continue
            d = int(dst_segs[i, 0])
            if r:
                dst[d:d + len(r)] = torch.frombuffer(bytearray(r), dtype=torch.uint8)
            out_sizes[i] = len(r)


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zstandard_amd
    import zstandard_amd.sharded as sh
    from tests.corpus import Corpus
    c = Corpus()
    items = [c.frame_bytes(i)[: 500 + 3100 * i] for i in range(11)] + [b"", b"x" * 40000, b"tail"]
    res = sh.multi_compress_to_buffer(items, level=3, gather=True, ctx_factory=_CpuCtx)
    full = res.to_buffer(zstandard_amd)                                  # one BufferWithSegments holding EVERY frame, on every rank
    frames = [full[i].tobytes() for i in range(len(full))]
    local = res.to_buffer(zstandard_amd, gathered=False)
    back = sh.multi_decompress_to_buffer(frames, [len(x) for x in items], gather=True, ctx_factory=_CpuCtx)
    outs = [bytes(back.full_arena[o:o + n].numpy()) for o, n in back.global_segments()]
    err = None
    try:
        sh.multi_decompress_to_buffer(frames[:5] + [frames[5][:-3] + b"zzz"] + frames[6:], [len(x) for x in items], ctx_factory=_CpuCtx)
    except zstandard_amd.ZstdError as e:
        err = str(e)
    # fewer items than ranks, and none at all: every rank still takes part in every collective (IndexError + hang before)
    small = []
    for sub in (items[3:4], []):
        r1 = sh.multi_compress_to_buffer(sub, level=3, gather=True, ctx_factory=_CpuCtx)
        f1 = r1.to_buffer(zstandard_amd)
        b1 = sh.multi_decompress_to_buffer([f1[i].tobytes() for i in range(len(f1))], [len(x) for x in sub], gather=True, ctx_factory=_CpuCtx)
        small.append((len(f1), [bytes(b1.full_arena[o:o + n].numpy()) for o, n in b1.global_segments()] == sub, r1.bounds))
    # a frame that decodes to fewer bytes than announced: reported like an item error on every rank, never a silently misaligned arena
    short_err = None
    try:
        sh.multi_decompress_to_buffer(frames, [len(x) + (7 if i == 2 else 0) for i, x in enumerate(items)], ctx_factory=_CpuCtx)
    except zstandard_amd.ZstdError as e:
        short_err = str(e)
    # every rank brings its OWN shard (uneven: rank 0 three items, rank 1 one -- and then none): compress_shard / decompress_shard
    shard_ok = []
    for mine in ((items[:3] if rank == 0 else items[5:6]), (items[:2] if rank == 0 else [])):
        ctx = _CpuCtx()
        arena = torch.frombuffer(bytearray(b"".join(mine) or b"\0"), dtype=torch.uint8)
        lens = torch.tensor([len(x) for x in mine], dtype=torch.int64)
        segs = torch.stack([torch.cumsum(lens, 0) - lens, lens], dim=1) if mine else torch.zeros((0, 2), dtype=torch.int64)
        rs = sh.compress_shard(ctx, arena, segs, gather=True)
        gs = rs.global_segments()
        allf = [bytes(rs.full_arena[o:o + n].numpy()) for o, n in gs]
        rd = sh.decompress_shard(ctx, rs.arena, rs.segs, lens, gather=True)
        back2 = [bytes(rd.full_arena[o:o + n].numpy()) for o, n in rd.global_segments()]
        shard_ok.append((rs.bounds, len(allf), [ctx.z.decompress(f, 1 << 20) for f in allf] == back2, back2[rs.lo:rs.hi] == list(mine)))
    q.put((rank, res.bounds, [len(f) for f in frames], outs == items, len(local), res.sizes.tolist(), err,
           [res.local_item(i) == frames[i] for i in range(res.lo, res.hi)], small, short_err, shard_ok))
    dist.destroy_process_group()


def test_sharded_calls_world2_full_control_flow(oracle):
    from tests import reflib
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=150) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, b0, l0, ok0, n0, s0, e0, loc0, sm0, se0, sh0), (r1, b1, l1, ok1, n1, s1, e1, loc1, sm1, se1, sh1) = res
    assert sm0 == sm1 and sm0[0][0] == 1 and sm0[0][1] and sm0[1][0] == 0 and sm0[1][1]      # one item / no items over two ranks
    assert len(sm0[0][2]) == 2 and sm0[0][2][1] == (1, 1) and sm0[1][2] == [(0, 0), (0, 0)]
    assert se0 == se1 and se0 is not None and "item 2" in se0 and "expected" in se0
    assert sh0 == sh1 and sh0[0][0] == [(0, 3), (3, 4)] and sh0[0][1] == 4 and sh0[0][2]
    assert sh0[1][0] == [(0, 2), (2, 2)] and sh0[1][1] == 2 and sh0[1][2]
    assert b0 == b1 and b0[0][0] == 0 and b0[0][1] == b0[1][0] and b0[1][1] == 14        # same contiguous cover on both ranks
    assert l0 == l1 == s0 == s1                                                           # every rank holds every frame + the global table
    assert ok0 and ok1                                                                    # round trip through both sharded calls
    assert n0 == b0[0][1] - b0[0][0] and n1 == b0[1][1] - b0[1][0]                        # the un-gathered buffer holds the rank's items
    assert all(loc0) and all(loc1)
    assert e0 == e1 and e0 is not None and "item 5" in e0                                 # both ranks raise the SAME first failing item
    # and the frames are libzstd's
    from tests.corpus import Corpus
    enc = reflib.RefZstd() if reflib.have_ref() else oracle
    assert l0[3] == len(enc.compress(Corpus().frame_bytes(3)[: 500 + 3100 * 3], level=3))


# ------------------------------------------------------------------------------------------------------------------------------
# the launch path of bench.py at N = 2: `python bench.py --gpus 2` must start its own ranks, and the driver's
# `python -m torch.distributed.run ... bench.py --gpus 2` must keep working. --dry-launch runs everything up to "process group formed,
# partition agreed" (gloo here: no GPU) and rank 0 prints the one JSON line.
def _dry_line(cmd):
    import json
    import subprocess
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["CUDA_VISIBLE_DEVICES"] = ""; env["HIP_VISIBLE_DEVICES"] = ""          # the CPU launch path, whatever the box has
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln[ln.index("{"):] for ln in p.stdout.splitlines() if '"dry_launch"' in ln]
    assert len(lines) == 1, p.stdout                                          # rank 0 alone speaks
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    line = _dry_line([sys.executable, "bench.py", "--gpus", "2", "--dry-launch", "--frames", "4096"])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert line["partition"] == [[0, 4096], [4096, 8192]]
    assert 1 <= line["host_threads_per_rank"] <= max(1, (os.cpu_count() or 1) // 2)


def test_bench_under_the_drivers_launcher():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    line = _dry_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", str(port), "bench.py", "--gpus", "2", "--dry-launch", "--frames", "65536"])
    assert line["n_gpus"] == 2 and line["partition"] == [[0, 65536], [65536, 131072]]
