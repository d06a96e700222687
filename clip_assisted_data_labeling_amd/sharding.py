"""Image sharding across the GPUs of one node and the single collective of the path.

The reference is single-process. Images are independent, so rank r of W takes a
contiguous block of the (sorted) image list, encodes + scores it with no inter-GPU traffic, and the
per-image results are collected ONCE (RCCL over xGMI when the tensors are on GPUs; gloo on CPU in the
tests): either gathered onto one rank (`dst=<rank>`, what BASELINE.json's north_star asks for: "only an
RCCL gather ... to collect embeddings/scores") or all-gathered onto every rank (`dst=None`).

Memory: the only allocation is the result itself, `[n_total, ...]`, on the ranks that receive it -- every
block is received straight into its final rows (no padded staging copy, no concatenation), so a rank
holds at most its own block + 1 x the result.
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


def gather_rows(local: torch.Tensor, n_total: int, group=None, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Collects the row blocks laid out by `shard_bounds` into the full `[n_total, ...]` tensor.

    dst=None: every rank receives the full tensor (all-gather).  Equal shards are one `all_gather_into_tensor` straight
              into the result; ragged shards (n_total % W != 0) are W broadcasts, each into its final rows.
    dst=r:    only rank r allocates and receives the result (point-to-point receives into the final rows); the other
              ranks send their block and get None.
    A single process returns `local` itself."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError(f"single process holds {local.shape[0]} rows, expected {n_total}")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard is {hi - lo}")
    if dst is not None and not (0 <= dst < world):
        raise ValueError(f"dst {dst} outside the group of {world}")
    local = local.contiguous()
    tail = tuple(local.shape[1:])

    def peer(r: int) -> int:                      # group rank -> global rank (send/recv/broadcast take global ranks)
        return dist.get_global_rank(group, r) if group is not None else r

    if dst is None:
        out = local.new_empty((n_total,) + tail)
        if n_total % world == 0:
            dist.all_gather_into_tensor(out, local, group=group)
            return out
        out[lo:hi].copy_(local)
        for r in range(world):
            rlo, rhi = shard_bounds(n_total, r, world)
            if rhi > rlo:
                dist.broadcast(out[rlo:rhi], src=peer(r), group=group)
        return out
    if rank != dst:
        if hi > lo:
            dist.send(local, dst=peer(dst), group=group)
        return None
    out = local.new_empty((n_total,) + tail)
    out[lo:hi].copy_(local)
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, r, world)
        if r != dst and rhi > rlo:
            dist.recv(out[rlo:rhi], src=peer(r), group=group)
    return out
