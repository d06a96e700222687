"""Image sharding across the GPUs of one node and the single collective of the path.

The reference is single-process. Images are independent, so rank r of W takes a
contiguous block of the (sorted) image list, encodes + scores it with no inter-GPU traffic, and the
per-image results are collected with one all-gather (RCCL over xGMI when the tensors are on GPUs;
gloo on CPU in the tests). Uneven tails are padded to the largest shard and trimmed after the gather.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of rank `rank`: the first n % W ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_bounds(len(items), rank, world)
    return list(items[lo:hi])


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather row blocks laid out by shard_bounds into the full [n_total, ...] tensor on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError(f"single process holds {local.shape[0]} rows, expected {n_total}")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard is {hi - lo}")
    longest = -(-n_total // world)
    padded = local.new_zeros((longest,) + tuple(local.shape[1:]))
    padded[: hi - lo] = local
    out = local.new_empty((world * longest,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    parts = []
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, r, world)
        parts.append(out[r * longest: r * longest + (rhi - rlo)])
    return torch.cat(parts, 0)
