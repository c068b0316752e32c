"""Batch sharding of independent MSAs over ranks (one process per GPU).  The path has no
per-step exchange: inputs are split contiguously, every rank runs its own rollouts, and
only the merge lists (int32 [B_local, T-1, 2]) are gathered at the end."""
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
