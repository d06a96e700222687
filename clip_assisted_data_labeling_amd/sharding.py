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
              into the result; ragged shards (n_total % W != 0) are one batch of point-to-point sends / receives posted
              together, each block received into its final rows.
    dst=r:    only rank r allocates and receives the result: the receives from ALL peers are posted together (one RCCL
              group call, every xGMI link busy at once) into the final rows; the other ranks send their block and get None.
    A single process WITHOUT a process group returns `local` itself; a group of one rank (a launcher with one GPU) takes the
    collective path like any other group -- one `all_gather_into_tensor` through the backend into a fresh result.

    Every call on a group starts with one tiny all-rank collective (`_warm_group`): the exchanges below are subsets of
    point-to-point operations, and on NCCL / RCCL a lazily initialised group (no `device_id=` at `init_process_group`) requires
    EVERY rank to take part in its first collective -- ranks with an empty shard post nothing in the ragged and `dst=` forms."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != n_total:
            raise ValueError(f"single process holds {local.shape[0]} rows, expected {n_total}")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    _warm_group(group, local.device)
    if world == 1:
        if local.shape[0] != n_total:
            raise ValueError(f"single process holds {local.shape[0]} rows, expected {n_total}")
        out = torch.empty_like(local, memory_format=torch.contiguous_format)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
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
        # ragged shards: every rank posts its W - 1 sends and W - 1 receives AT ONCE (one RCCL group call), each block
        # received straight into its final rows -- all xGMI links carry traffic together, instead of W broadcasts in turn
        out[lo:hi].copy_(local)
        ops = []
        for r in range(world):
            rlo, rhi = shard_bounds(n_total, r, world)
            if r == rank:
                continue
            if rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], peer(r), group))
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, local, peer(r), group))
        _run_p2p(ops)
        return out
    if rank != dst:
        if hi > lo:
            _run_p2p([dist.P2POp(dist.isend, local, peer(dst), group)])
        return None
    # true gather: ALL peers' receives are posted together (SURVEY section 5.8 prices the gather of configs[3] at ~10 ms
    # BECAUSE the 7 peers write at the same time, one xGMI link each; receiving them in turn used one link at a time)
    out = local.new_empty((n_total,) + tail)
    out[lo:hi].copy_(local)
    ops = []
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, r, world)
        if r != dst and rhi > rlo:
            ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], peer(r), group))
    _run_p2p(ops)
    return out


def _warm_group(group, device) -> None:
    """One all-reduce of a single element on `device` in front of every exchange (tens of microseconds; gather_rows runs twice
    per job): every rank calls gather_rows, so every rank takes part -- after it the backend's communicator exists on all of
    them and subset point-to-point batches are well defined, whether or not the group was initialised eagerly."""
    dist.all_reduce(torch.zeros(1, device=device), group=group)


def _run_p2p(ops) -> None:
    """Posts a list of P2POp together (`batch_isend_irecv`: one ncclGroupStart / End on RCCL, plain isend / irecv on
    gloo) and waits for all of them."""
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()
