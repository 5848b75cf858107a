"""Sharded batch calls: ``multi_compress_to_buffer`` / ``multi_decompress_to_buffer`` across the GPUs of a node, one process per GPU.

The reference fans a batch out INSIDE the call: the item list is cut into contiguous ranges of (almost) equal input bytes, one per
worker thread, each worker fills its own destination buffers, and the results are collected into one BufferWithSegmentsCollection
(c-ext/compressor.c:1127-1216 + :1340-1503, c-ext/decompressor.c:1237-1320 + :1459-1710). Here a worker is a rank (a GPU):

    every rank calls with the SAME item list  ->  parallel.my_shard (the reference's partition rule)  ->  this rank's range through
    its DeviceBatchContext (HBM-resident, the kernels of csrc/)  ->  ShardResult: this rank's output arena + segment table, and the
    GLOBAL table of output sizes (an all-gather of a few hundred KiB)  ->  optionally `gather=True`: the payload all-gatherv over
    RCCL / xGMI, after which every rank holds the complete output arena in item order (one BufferWithSegments for the whole call).

The payload gather is a sequence of N broadcasts into the slices of ONE preallocated arena (rank r is the root of slice r); all are
issued asynchronously and waited for together, which over RCCL is N concurrent ring broadcasts on the 7 xGMI links of every GPU.
Results that the caller consumes where they were produced (the common case: the decompressed data feeds a GPU pipeline) need no
gather at all, which is why it is optional and off by default.

`ctx_factory` exists so that the control flow -- partition, local call, size exchange, payload gather, reassembly -- can run in the
world-size-2 CPU test (gloo) with a stand-in context; the default is the real DeviceBatchContext and needs a GPU (no CPU fallback).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import parallel


class ShardResult:
    """What one rank holds after a sharded call.

    bounds        [(lo, hi)] per rank: the item range each rank processed (the reference's partition by input bytes)
    arena, segs   this rank's output: uint8 arena on its device, int64 [hi - lo, 2] (offset, length) into it
    sizes         int64 [n] on the host: output length of EVERY item (global order), identical on all ranks
    status        int32 [hi - lo]: per-item zstd error code of this rank's items (0 = ok)
    full_arena    after gather: uint8 arena with every item's output in item order (on this rank's device), else None
    """

    def __init__(self, rank, bounds, arena, segs, sizes, status):
        self.rank, self.bounds, self.arena, self.segs, self.sizes, self.status = rank, bounds, arena, segs, sizes, status
        self.full_arena = None

    @property
    def lo(self):
        return self.bounds[self.rank][0]

    @property
    def hi(self):
        return self.bounds[self.rank][1]

    def local_item(self, i):
        """bytes of global item i if this rank produced it"""
        assert self.lo <= i < self.hi
        off, ln = (int(v) for v in self.segs[i - self.lo])
        return bytes(self.arena[off:off + ln].cpu().numpy())

    def global_segments(self):
        """int64 [n, 2] (offset, length) of every item inside the gathered arena (item order, densely packed)"""
        offs = np.zeros(len(self.sizes), dtype=np.int64)
        if len(self.sizes):
            offs[1:] = np.cumsum(self.sizes)[:-1]
        return np.stack([offs, np.asarray(self.sizes, dtype=np.int64)], axis=1)

    def to_buffer(self, zstd_module, gathered=None):
        """a host-side BufferWithSegments: of the gathered arena (every item) when the payload was gathered, else of this rank's items"""
        use_full = self.full_arena is not None if gathered is None else gathered
        if use_full:
            assert self.full_arena is not None, "call with gather=True first"
            data, sg = self.full_arena.cpu().numpy(), self.global_segments()
        else:
            data, sg = self.arena.cpu().numpy(), np.ascontiguousarray(self.segs.cpu().numpy())
        return zstd_module.BufferWithSegments(data.tobytes(), sg.astype("<u8").tobytes())


def _default_ctx_factory(**kw):
    from .device import DeviceBatchContext
    return DeviceBatchContext(**kw)


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def _pack(items, lo, hi, dev):
    """this rank's items back to back in one device arena + their segment table"""
    lens = np.fromiter((len(items[i]) for i in range(lo, hi)), dtype=np.int64, count=hi - lo)
    offs = np.zeros(hi - lo, dtype=np.int64)
    if hi > lo:
        offs[1:] = np.cumsum(lens)[:-1]
    host = np.frombuffer(b"".join(bytes(items[i]) for i in range(lo, hi)), dtype=np.uint8)
    arena = torch.from_numpy(host.copy() if host.size else np.zeros(1, dtype=np.uint8)).to(dev)
    segs = torch.from_numpy(np.stack([offs, lens], axis=1) if hi > lo else np.zeros((0, 2), dtype=np.int64)).to(dev)
    return arena, segs, lens


def _exchange_sizes(local_sizes, group):
    """all-gather of the per-item output sizes -> int64 numpy [n] in item order (ranks may hold different counts)"""
    per_rank = parallel.gather_segment_table(local_sizes, group=group)
    return np.concatenate([t.cpu().numpy() for t in per_rank]) if per_rank else np.zeros(0, dtype=np.int64)


def allgatherv_payload(result, group=None):
    """the payload all-gatherv: every rank ends up with every item's output, in item order, in ONE arena on its own device
    (result.full_arena). Rank r's slice of that arena is the concatenation of its items, which is exactly its local arena when that
    is dense (what both directions below produce after compaction), so the exchange is one broadcast per rank, all in flight together."""
    world = dist.get_world_size(group)
    sizes = result.sizes
    starts = np.zeros(len(sizes) + 1, dtype=np.int64)
    starts[1:] = np.cumsum(sizes)
    full = torch.empty(max(int(starts[-1]), 1), dtype=torch.uint8, device=result.arena.device)
    works = []
    for r in range(world):
        lo, hi = result.bounds[r]
        a, b = int(starts[lo]), int(starts[hi])
        if b == a:
            continue
        view = full[a:b]
        if r == result.rank:
            view.copy_(result.arena[: b - a])
        src = dist.get_global_rank(group, r) if group is not None else r
        works.append(dist.broadcast(view, src=src, group=group, async_op=True))
    for w in works:
        w.wait()
    result.full_arena = full[: int(starts[-1])]
    return result


def _compact(arena, segs, out_sizes, dev, step_bytes=256 << 20):
    """dense copy of the slots' valid prefixes (device-side gather through index tensors, at most `step_bytes` of output at a time): the
    compressed direction's slots are compressBound-sized, what is handed on or gathered is only the frames"""
    n = segs.shape[0]
    lens = out_sizes.to(torch.int64)
    offs = torch.cumsum(lens, 0) - lens
    total = int(lens.sum().item()) if n else 0
    dense_segs = torch.stack([offs, lens], dim=1)
    if total == 0:
        return torch.zeros(1, dtype=torch.uint8, device=dev), dense_segs
    dense = torch.empty(total, dtype=torch.uint8, device=dev)
    ends = (offs + lens).cpu().numpy()
    lo = 0
    while lo < n:
        hi = int(np.searchsorted(ends, ends[lo] - int(lens[lo].item()) + step_bytes, side="right"))
        hi = max(hi, lo + 1)
        l = lens[lo:hi]
        cnt = int(l.sum().item())
        if cnt:
            base = int(offs[lo].item())
            item_of = torch.repeat_interleave(torch.arange(hi - lo, device=dev), l)
            pos = torch.arange(cnt, device=dev) - (offs[lo:hi] - base)[item_of] + segs[lo:hi, 0][item_of]
            dense[base:base + cnt] = arena[pos]
        lo = hi
    return dense, dense_segs


def multi_decompress_to_buffer(frames, decompressed_sizes, dict_data=None, gather=False, group=None, ctx=None, ctx_factory=None, **ctx_kw):
    """Sharded ZstdDecompressor.multi_decompress_to_buffer. `frames`: the whole call's frames (any sequence of bytes-like objects; a rank
    only touches its own range), `decompressed_sizes`: their content sizes (sequence of ints; the reference reads them from the
    frame headers or takes them as its `decompressed_sizes` argument). Returns a ShardResult; raises if any local item fails."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = parallel.partition_by_bytes([len(f) for f in frames], world)
    lo, hi = bounds[rank]
    own_ctx = ctx is None
    ctx = ctx if ctx is not None else (ctx_factory or _default_ctx_factory)(dict_data=dict_data, **ctx_kw)
    dev = getattr(ctx, "device", None) or _device()
    src, src_segs, _ = _pack(frames, lo, hi, dev)
    want = np.asarray([int(decompressed_sizes[i]) for i in range(lo, hi)], dtype=np.int64)
    offs = np.zeros(hi - lo, dtype=np.int64)
    if hi > lo:
        offs[1:] = np.cumsum(want)[:-1]
    dst = torch.zeros(max(int(want.sum()), 1), dtype=torch.uint8, device=dev)
    dst_segs = torch.from_numpy(np.stack([offs, want], axis=1) if hi > lo else np.zeros((0, 2), dtype=np.int64)).to(dev)
    out_sizes = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
    status = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
    if hi > lo:
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    _raise_on_error(status, lo, "decompressing", group)
    sizes = _exchange_sizes(out_sizes, group)
    res = ShardResult(rank, bounds, dst, torch.stack([dst_segs[:, 0], out_sizes], dim=1) if hi > lo else dst_segs, sizes, status)
    if own_ctx and hasattr(ctx, "close"):
        ctx.close()
    return allgatherv_payload(res, group) if gather else res


def multi_compress_to_buffer(items, level=3, dict_data=None, gather=False, group=None, ctx=None, ctx_factory=None, **ctx_kw):
    """Sharded ZstdCompressor.multi_compress_to_buffer: every frame is what libzstd produces for that item at that level, whichever rank
    made it. Returns a ShardResult whose arena holds this rank's frames densely packed."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = parallel.partition_by_bytes([len(x) for x in items], world)
    lo, hi = bounds[rank]
    own_ctx = ctx is None
    ctx = ctx if ctx is not None else (ctx_factory or _default_ctx_factory)(dict_data=dict_data, level=level, **ctx_kw)
    dev = getattr(ctx, "device", None) or _device()
    src, src_segs, lens = _pack(items, lo, hi, dev)
    bound = [(int(n) + (int(n) >> 8) + 64 + 15) & ~15 for n in lens]
    offs = np.zeros(hi - lo, dtype=np.int64)
    if hi > lo:
        offs[1:] = np.cumsum(bound)[:-1]
    dst = torch.zeros(max(int(sum(bound)), 1), dtype=torch.uint8, device=dev)
    dst_segs = torch.from_numpy(np.stack([offs, np.asarray(bound, dtype=np.int64)], axis=1) if hi > lo else np.zeros((0, 2), dtype=np.int64)).to(dev)
    out_sizes = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
    status = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
    if hi > lo:
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    _raise_on_error(status, lo, "compressing", group)
    arena, segs = _compact(dst, dst_segs, out_sizes, dev)
    sizes = _exchange_sizes(out_sizes, group)
    res = ShardResult(rank, bounds, arena, segs, sizes, status)
    if own_ctx and hasattr(ctx, "close"):
        ctx.close()
    return allgatherv_payload(res, group) if gather else res


def _raise_on_error(status, lo, what, group):
    """the reference reports the first failing item of the whole call (compressor.c:1290-1310, decompressor.c:1400-1430): agree on it
    across ranks so that every rank raises the same error instead of some hanging in the next collective"""
    from .backend_hip import ZstdError
    bad = torch.nonzero(status).flatten()
    first = int(bad[0].item()) + lo if bad.numel() else -1
    code = int(status[bad[0]].item()) if bad.numel() else 0
    t = torch.tensor([first if first >= 0 else (1 << 62), code], dtype=torch.int64, device=status.device)
    all_t = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(all_t, t, group=group)
    worst = min(all_t, key=lambda v: int(v[0].item()))
    if int(worst[0].item()) != (1 << 62):
        from . import _lib
        raise ZstdError("error %s item %d: %s" % (what, int(worst[0].item()), _lib.error_name(int(worst[1].item()))))
