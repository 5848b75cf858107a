"""Sharded batch calls: ``multi_compress_to_buffer`` / ``multi_decompress_to_buffer`` across the GPUs of a node, one process per GPU.

The reference fans a batch out INSIDE the call: the item list is cut into contiguous ranges of (almost) equal input bytes, one per
worker thread, each worker fills its own destination buffers, and the results are collected into one BufferWithSegmentsCollection
(c-ext/compressor.c:1127-1216 + :1340-1503, c-ext/decompressor.c:1237-1320 + :1459-1710). Here a worker is a rank (a GPU):

    every rank calls with the SAME item list  ->  parallel.my_shard (the reference's partition rule)  ->  this rank's range through
    its DeviceBatchContext (HBM-resident, the kernels of csrc/)  ->  ShardResult: this rank's output arena + segment table, and the
    GLOBAL table of output sizes (an all-gather of a few hundred KiB)  ->  optionally `gather=True`: the payload all-gatherv over
    RCCL / xGMI, after which every rank holds the complete output arena in item order (one BufferWithSegments for the whole call).

The payload gather is ONE collective: every rank's dense slice, padded to the longest slice, through all_gather_into_tensor (over RCCL
a single ring all-gather that keeps all 7 xGMI links of every GPU busy; the partition is balanced by bytes, so the padding is small).
The gathered arena keeps the padded layout -- a BufferWithSegments may have gaps -- and global_segments() addresses it, so no second
copy is made. Results that the caller consumes where they were produced (the common case: the decompressed data feeds a GPU pipeline)
need no gather at all, which is why it is optional and off by default.

A rank needs the SIZES of every item (the partition rule walks them) but the BYTES of its own range only: `items` may be any object
indexable by global item number over the rank's own range (e.g. a dict, or a lazy loader) when `sizes=` is given. Ranks that already
hold their shard in HBM (bench.py's configs[4]: every rank generates its own) call compress_shard / decompress_shard instead.

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
        self.slice_stride = 0           # after gather: rank r's items start at r * slice_stride inside full_arena

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
        """int64 [n, 2] (offset, length) of every item inside the gathered arena, item order: rank r's items lie back to back from
        r * slice_stride on (slices are padded to the longest one; a BufferWithSegments may have gaps)"""
        sizes = np.asarray(self.sizes, dtype=np.int64)
        offs = np.zeros(len(sizes), dtype=np.int64)
        for r, (lo, hi) in enumerate(self.bounds):
            if hi > lo:
                c = np.cumsum(sizes[lo:hi])
                offs[lo:hi] = r * self.slice_stride + c - sizes[lo:hi]
        return np.stack([offs, sizes], axis=1)

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
    """this rank's items back to back in one device arena + their segment table. The items are copied ONCE, straight into a pinned staging
    buffer (no intermediate joined bytes object), which then crosses the link in a single asynchronous copy."""
    lens = np.fromiter((len(items[i]) for i in range(lo, hi)), dtype=np.int64, count=hi - lo)
    offs = np.zeros(hi - lo, dtype=np.int64)
    if hi > lo:
        offs[1:] = np.cumsum(lens)[:-1]
    total = int(lens.sum()) if hi > lo else 0
    stage = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=dev.type == "cuda")
    view = stage.numpy()
    for k, i in enumerate(range(lo, hi)):
        if lens[k]:
            view[offs[k]:offs[k] + lens[k]] = np.frombuffer(items[i], dtype=np.uint8)
    arena = stage.to(dev, non_blocking=True)
    segs = torch.from_numpy(np.stack([offs, lens], axis=1) if hi > lo else np.zeros((0, 2), dtype=np.int64)).to(dev)
    if dev.type == "cuda":
        torch.cuda.current_stream().synchronize()           # the pinned staging buffer may go once the copy has landed
    return arena, segs, lens


def _exchange_sizes(local_sizes, group):
    """all-gather of the per-item output sizes -> int64 numpy [n] in item order (ranks may hold different counts)"""
    per_rank = parallel.gather_segment_table(local_sizes, group=group)
    return np.concatenate([t.cpu().numpy() for t in per_rank]) if per_rank else np.zeros(0, dtype=np.int64)


def allgatherv_payload(result, group=None):
    """the payload all-gatherv: every rank ends up with every item's output in ONE arena on its own device (result.full_arena; item i at
    result.global_segments()[i]). Rank r's slice is the concatenation of its items -- exactly its local arena, which both directions
    below leave dense -- padded to the longest slice so that the exchange is a single all_gather_into_tensor."""
    world = dist.get_world_size(group)
    sizes = np.asarray(result.sizes, dtype=np.int64)
    slice_bytes = [int(sizes[lo:hi].sum()) for lo, hi in result.bounds]
    stride = (max(slice_bytes + [1]) + 255) & ~255
    mine = slice_bytes[result.rank]
    dev = result.arena.device
    send = result.arena
    if send.numel() < stride:                                # (the local arena is rarely the longest: pad it once)
        send = torch.zeros(stride, dtype=torch.uint8, device=dev)
        send[:mine] = result.arena[:mine]
    full = torch.empty(world * stride, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(full, send[:stride], group=group)
    result.full_arena, result.slice_stride = full, stride
    return result


def _lib_compact(arena, segs, out_sizes, status, offs, dense):
    """the library's device compaction (one wave per frame): needs the GPU library -- there is no host path"""
    from . import _lib
    rc = _lib.lib().zhip_compact_device(arena.data_ptr(), segs.data_ptr(), out_sizes.data_ptr(), status.data_ptr(), offs.data_ptr(),
                                        segs.shape[0], dense.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc:
        from .backend_hip import ZstdError
        raise ZstdError("HIP backend failure: %s" % _lib.last_error())


def _compact(arena, segs, out_sizes, status, dev, compactor=None):
    """dense copy of the slots' valid prefixes: the compress direction's slots are compressBound-sized, what is handed on or gathered is
    only the frames. The copy is the library's zhip_compact_device kernel (include/zstd_hip.h); the world-size-2 CPU test's stand-in
    context brings its own `compact`."""
    n = segs.shape[0]
    lens = torch.where(status != 0, torch.zeros_like(out_sizes), out_sizes).to(torch.int64)
    offs = (torch.cumsum(lens, 0) - lens).contiguous()
    total = int(lens.sum().item()) if n else 0
    dense_segs = torch.stack([offs, lens], dim=1)
    dense = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
    if total:
        (compactor or _lib_compact)(arena, segs.contiguous(), out_sizes.contiguous(), status.contiguous(), offs, dense)
    return dense, dense_segs


def _bounds_from_counts(count, group):
    """item ranges of every rank when each rank brings its own shard: ranks in order, (lo, hi) from the all-gathered counts"""
    world = dist.get_world_size(group)
    t = torch.tensor([count], dtype=torch.int64, device=_cdev(group))
    all_t = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(all_t, t, group=group)
    counts = [int(v.item()) for v in all_t]
    ends = np.cumsum(counts)
    return [(int(e - c), int(e)) for e, c in zip(ends, counts)]


def _cdev(group):
    """device of the small metadata tensors of the collectives: the current GPU under RCCL, the host under gloo"""
    return _device() if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _hint(ctx, largest):
    """tell the context how large this rank's largest (uncompressed) item is: items above one block take the kernels' several-block modes
    (zhip_ctx_set_size_hint); stand-in contexts of the CPU tests need not know the call"""
    if hasattr(ctx, "set_size_hint"):
        ctx.set_size_hint(int(largest))


def compress_shard(ctx, src, src_segs, gather=False, group=None):
    """Every rank compresses the shard IT holds (src / src_segs: device arena + int64 [k, 2] segments; ranks may hold different
    counts, none included); global item order = rank order. Returns a ShardResult like multi_compress_to_buffer. Nothing but the
    per-item sizes (and, with gather=True, the frames) crosses ranks; no rank ever sees another rank's input."""
    rank = dist.get_rank(group)
    dev = src.device
    k = src_segs.shape[0]
    bounds = _bounds_from_counts(k, group)
    lens = src_segs[:, 1]
    bound = ((lens + (lens >> 8) + 64 + 15) & ~15) if k else lens
    offs = torch.cumsum(bound, 0) - bound if k else bound
    dst = torch.zeros(max(int(bound.sum().item()) if k else 0, 1), dtype=torch.uint8, device=dev)
    dst_segs = torch.stack([offs, bound], dim=1).contiguous() if k else torch.zeros((0, 2), dtype=torch.int64, device=dev)
    out_sizes = torch.zeros(k, dtype=torch.int64, device=dev)
    status = torch.zeros(k, dtype=torch.int32, device=dev)
    if k:
        _hint(ctx, lens.max().item())
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    _raise_on_error(status, bounds[rank][0], "compressing", group)
    arena, segs = _compact(dst, dst_segs, out_sizes, status, dev, getattr(ctx, "compact", None))
    res = ShardResult(rank, bounds, arena, segs, _exchange_sizes(out_sizes, group), status)
    return allgatherv_payload(res, group) if gather else res


def decompress_shard(ctx, src, src_segs, sizes, gather=False, group=None):
    """The opposite direction on the shard this rank holds; `sizes`: int64 tensor [k] of the frames' content sizes."""
    rank = dist.get_rank(group)
    dev = src.device
    k = src_segs.shape[0]
    bounds = _bounds_from_counts(k, group)
    want = sizes.to(torch.int64)
    offs = torch.cumsum(want, 0) - want if k else want
    dst = torch.zeros(max(int(want.sum().item()) if k else 0, 1), dtype=torch.uint8, device=dev)
    dst_segs = torch.stack([offs, want], dim=1).contiguous() if k else torch.zeros((0, 2), dtype=torch.int64, device=dev)
    out_sizes = torch.zeros(k, dtype=torch.int64, device=dev)
    status = torch.zeros(k, dtype=torch.int32, device=dev)
    if k:
        _hint(ctx, want.max().item())
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    _raise_on_error(status, bounds[rank][0], "decompressing", group, out_sizes, want)
    res = ShardResult(rank, bounds, dst, torch.stack([offs, out_sizes], dim=1) if k else dst_segs, _exchange_sizes(out_sizes, group), status)
    return allgatherv_payload(res, group) if gather else res


def multi_decompress_to_buffer(frames, decompressed_sizes, dict_data=None, gather=False, group=None, ctx=None, ctx_factory=None, sizes=None, **ctx_kw):
    """Sharded ZstdDecompressor.multi_decompress_to_buffer. `frames`: the whole call's frames (any sequence of bytes-like objects; a rank
    only touches its own range), `decompressed_sizes`: their content sizes (sequence of ints; the reference reads them from the
    frame headers or takes them as its `decompressed_sizes` argument). Returns a ShardResult; raises if any local item fails."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = parallel.partition_by_bytes(sizes if sizes is not None else [len(f) for f in frames], world)
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
        _hint(ctx, want.max())
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    # a frame that produces fewer bytes than announced would leave a hole in the dense arena the global table assumes: the host API's
    # "decompressed N bytes; expected M" (c-ext/decompressor.c:1131-1140), agreed on across ranks like any other item error
    _raise_on_error(status, lo, "decompressing", group, out_sizes, torch.from_numpy(want).to(dev))
    all_sizes = _exchange_sizes(out_sizes, group)
    res = ShardResult(rank, bounds, dst, torch.stack([dst_segs[:, 0], out_sizes], dim=1) if hi > lo else dst_segs, all_sizes, status)
    if own_ctx and hasattr(ctx, "close"):
        ctx.close()
    return allgatherv_payload(res, group) if gather else res


def multi_compress_to_buffer(items, level=3, dict_data=None, gather=False, group=None, ctx=None, ctx_factory=None, sizes=None, **ctx_kw):
    """Sharded ZstdCompressor.multi_compress_to_buffer: every frame is what libzstd produces for that item at that level, whichever rank
    made it. Returns a ShardResult whose arena holds this rank's frames densely packed."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = parallel.partition_by_bytes(sizes if sizes is not None else [len(x) for x in items], world)
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
        _hint(ctx, max(int(n) for n in lens))
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    _raise_on_error(status, lo, "compressing", group)
    arena, segs = _compact(dst, dst_segs, out_sizes, status, dev, getattr(ctx, "compact", None))
    all_sizes = _exchange_sizes(out_sizes, group)
    res = ShardResult(rank, bounds, arena, segs, all_sizes, status)
    if own_ctx and hasattr(ctx, "close"):
        ctx.close()
    return allgatherv_payload(res, group) if gather else res


def _raise_on_error(status, lo, what, group, got=None, want=None):
    """the reference reports the first failing item of the whole call (compressor.c:1290-1310, decompressor.c:1400-1430): agree on it
    across ranks so that every rank raises the same error instead of some hanging in the next collective"""
    from .backend_hip import ZstdError
    flag = status != 0
    if got is not None and got.numel():
        flag = flag | (got != want)
    bad = torch.nonzero(flag).flatten()
    first = int(bad[0].item()) + lo if bad.numel() else -1
    code = int(status[bad[0]].item()) if bad.numel() else 0
    if bad.numel() and code == 0:
        code = -(1 + int(got[bad[0]].item()))                # size mismatch: carries the produced size
    t = torch.tensor([first if first >= 0 else (1 << 62), code], dtype=torch.int64, device=status.device)
    all_t = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(all_t, t, group=group)
    worst = min(all_t, key=lambda v: int(v[0].item()))
    if int(worst[0].item()) != (1 << 62):
        code = int(worst[1].item())
        if code < 0:
            raise ZstdError("error %s item %d: decompressed %d bytes; expected another size" % (what, int(worst[0].item()), -code - 1))
        from . import _lib
        raise ZstdError("error %s item %d: %s" % (what, int(worst[0].item()), _lib.error_name(code)))
