"""Frame sharding of the evaluation stream across GPUs.

One process per GPU (torch.distributed: "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).
The id-sorted item stream is cut into contiguous shards; THC/TPC need the heat-maps of the id-adjacent
items, so a shard is extended by a one-item halo on each interior side and the halo items are simply
re-computed locally (two extra forwards per rank instead of any heat-map exchange).  The only
collective is one all-gather of the per-item result rows (~290 bytes per item).  Parameters live on
every GPU: nothing is re-broadcast per call (the reference's DataParallel re-broadcasts 136 MB per
forward, ActiveLearning.py:233,277).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous balanced shard [lo, hi) of n items."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def halo_bounds(n: int, lo: int, hi: int, halo: int = 1):
    """Shard extended by the halo, clipped to the stream: (lo_ext, hi_ext, skip_front, skip_back)."""
    lo_e, hi_e = max(0, lo - halo), min(n, hi + halo)
    return lo_e, hi_e, lo - lo_e, hi_e - hi


def sharded_rows(n: int, score_fn, row_width: int, device, halo: int = 1) -> torch.Tensor:
    """Every rank scores its shard (+halo) with ``score_fn(lo_ext, hi_ext) -> (hi_ext-lo_ext, row_width)`` rows (any
    one dtype) on ``device``; returns the (n, row_width) result rows of the whole stream on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(n, rank, world)
    lo_e, hi_e, front, back = halo_bounds(n, lo, hi, halo)
    rows = score_fn(lo_e, hi_e)
    rows = rows[front:rows.shape[0] - back].contiguous()
    assert rows.shape == (hi - lo, row_width), (rows.shape, hi - lo, row_width)
    if world == 1:
        return rows
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    pad = max(h - l for l, h in sizes)
    buf = torch.zeros((pad, row_width), device=device, dtype=rows.dtype)
    buf[:hi - lo] = rows
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:h - l] for o, (l, h) in zip(out, sizes)], 0)


def allreduce_mean_(tensors, group=None):
    """Gradient averaging for the data-parallel fine-tune step: one flat fp32 bucket, one all-reduce
    (136 MB for SimplePose-R50), divided by the world size."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size()
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


def broadcast_buffers_(module, src: int = 0):
    """BatchNorm running statistics are computed per rank during a data-parallel fine-tune (like the replicas of
    nn.DataParallel); the copy that survives is rank ``src``'s."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src=src)


def is_main() -> bool:
    return not dist.is_initialized() or dist.get_rank() == 0


def world_rank():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)


def shared_seed() -> int:
    """One random seed agreed by all ranks (rank 0's): the ranks must shuffle the fine-tune set identically before
    each takes its slice of every mini-batch."""
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        seed = seed.to(dev)
        dist.broadcast(seed, src=0)
    return int(seed.item())


def broadcast_module_(module, src: int = 0):
    """Parameters and buffers of rank ``src`` to every rank (after a random initialisation or a rank-local fit, so
    that the replicas start identical — what nn.DataParallel's per-forward replicate does for the reference)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
        torch.autograd.graph.increment_version(t)
