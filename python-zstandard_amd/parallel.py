"""Multi-GPU sharding of the batch hot path: one process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).

Frames are independent, so the only "parallelism strategy" is the reference's own: a static contiguous partition of the
item list with (almost) equal input bytes per worker -- c-ext/compressor.c:1127-1216 / c-ext/decompressor.c:1237-1326 --
here with one worker per rank instead of one per pthread. No collective touches the payload: results stay resident in
each rank's HBM (the returned collection holds one BufferWithSegments per rank, which the API allows);
the exchange every call needs is an all-gather of the per-frame output sizes so every rank knows the global segment table.
The callable form of all this -- including the optional payload all-gatherv over RCCL -- is sharded.py; this module holds the
partition rule and the metadata exchange it is built from.
"""
import torch
import torch.distributed as dist


def partition_by_bytes(sizes, workers):
    """Greedy contiguous partition used by the reference's dispatcher: walk the items, cut when the running byte count
    reaches total/workers, the last worker takes the rest. Returns [(start, end)] with `end` exclusive, one per worker
    (trailing workers may be empty when there are fewer items than workers)."""
    n = len(sizes)
    asked = max(1, int(workers))
    workers = max(1, min(asked, n)) if n else 1           # the reference's clamp (compressor.c:1151): never more workers than items
    total = int(sum(sizes))
    per = total // workers
    bounds, start, acc, w = [], 0, 0, 0
    for i, s in enumerate(sizes):
        acc += int(s)
        if w < workers - 1 and acc >= per:
            bounds.append((start, i + 1))
            start, acc, w = i + 1, 0, w + 1
    bounds.append((start, n))
    while len(bounds) < asked:                             # one entry per ASKED worker: a rank without items still takes part in the collectives
        bounds.append((n, n))
    return bounds


def my_shard(sizes, rank=None, world=None, group=None):
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    bounds = partition_by_bytes(sizes, world)
    return bounds[rank] if rank < len(bounds) else (len(sizes), len(sizes))


def gather_segment_table(local_sizes, counts=None, group=None):
    """all-gather of per-frame output sizes (int64 tensor on this rank's device) -> list of per-rank size tensors.
    Ranks may hold different numbers of frames: counts are exchanged first, then padded sizes."""
    world = dist.get_world_size(group)
    dev = local_sizes.device
    n = torch.tensor([local_sizes.numel()], dtype=torch.int64, device=dev)
    all_n = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(all_n, n, group=group)
    counts = [int(t.item()) for t in all_n]
    m = max(counts) if counts else 0
    padded = torch.zeros(m, dtype=torch.int64, device=dev)
    padded[: local_sizes.numel()] = local_sizes
    gathered = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    return [g[:c] for g, c in zip(gathered, counts)]


def global_segments(per_rank_sizes):
    """(rank, local offset, length) triples of the sharded result in global frame order."""
    out = []
    for r, sizes in enumerate(per_rank_sizes):
        off = 0
        for s in sizes.tolist():
            out.append((r, off, int(s)))
            off += int(s)
    return out
