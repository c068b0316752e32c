"""Batch sharding of independent MSAs over ranks (one process per GPU).  The inference path has no
per-step exchange: inputs are split contiguously, every rank runs its own rollouts, and
only the merge lists (int32 [B_local, T-1, 2]) are gathered at the end.
The Finetune mode has ONE real exchange step per optimizer step: the episodes of an epoch are split over the ranks and
the gradients (425,857 floats, 1.7 MB) are summed by a single all-reduce of one flat bucket (allreduce_gradients)."""
from __future__ import annotations

import torch


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous, balanced split: the first (total % world) ranks get one extra item."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(packed: torch.Tensor, dist, src: int = 0) -> torch.Tensor:
    """Rank `src`'s packed weights to every rank (1.7 MB, once at start-up)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(packed, src=src)
    return packed


def gather_merges(local: torch.Tensor, total: int, dist):
    """all_gather of the per-rank merge lists, un-padded and concatenated in batch order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    cap = max(sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def allreduce_gradients(params, dist):
    """Sum of the gradients of `params` over the ranks, in place: every gradient is copied into ONE flat bucket (the whole
    model is 1.7 MB: one collective per optimizer step, not one per tensor -- on point-to-point xGMI a ring all-reduce
    is latency bound at this size), all-reduced (RCCL under "nccl"), and copied back.  Parameters without a gradient
    contribute zeros, so every rank reduces the same layout."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = [p for p in params if p.requires_grad]
    sizes = [p.numel() for p in params]
    ref = next(p for p in params)
    bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=ref.device)
    off = 0
    for p, n in zip(params, sizes):
        if p.grad is not None:
            bucket[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    off = 0
    for p, n in zip(params, sizes):
        g = bucket[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
