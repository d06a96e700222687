"""Whole-job runner for the embed + score path: N images sharded over the ranks of one node, streamed through the encoder
in batches, results kept on the device, ONE gather at the end (BASELINE.json configs[3]: "1M synthetic images sharded over
8 x MI355X with RCCL gather").

This is the loop of /root/reference/_1_embed_with_CLIP.py:100-170 (batch -> encode_image -> keep the features) with the
scoring of /root/reference/_5_predict_labels.py:133-135 fused behind it, minus the per-image `.pt` round trip: rank r of W owns
the contiguous block `sharding.shard_bounds(N, r, W)` of the image index range, encodes it `batch_images` at a time, writes
embeddings `[n_local, crops, E]` and scores `[n_local, out]` into two device tensors, and after its last batch the blocks
of all ranks are collected with `sharding.gather_rows` (one all_gather_into_tensor per tensor: RCCL over xGMI on GPUs, gloo
in the CPU tests).  No data-path collective before that.

The synthetic source generates a batch ON THE DEVICE from a counter-based generator seeded `base_seed + rank`
(torch's device generators are Philox counters): 1 M images x 4 crops of 3 x 224 x 224 would be 2.4 TB as fp32 and cannot be
staged, so every batch exists only while it is being encoded -- uint8 pixels, 0.31 GB per 512-image
batch, normalised inside the encoder's first kernel (CLIPENC_IN_U8).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .sharding import gather_rows, shard_bounds


def synthetic_u8_source(image_size: int, crops_per_image: int, base_seed: int, rank: int, device) -> Callable[[int, int], torch.Tensor]:
    """source(first_image, n_images) -> uint8 [n_images * crops, 3, R, R] generated on `device`.
    The stream of a rank depends only on (base_seed + rank) and on how many values were drawn before, i.e. on the batch
    sequence of that rank -- a re-run of the same job reproduces every batch."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(base_seed) + int(rank))

    def source(first_image: int, n_images: int) -> torch.Tensor:
        return torch.randint(0, 256, (n_images * crops_per_image, 3, image_size, image_size), generator=gen, device=device,
                             dtype=torch.uint8)
    return source


def run_embed_job(n_images: int, batch_images: int, crops_per_image: int, embed_dim: int, score_dim: int,
                  source: Callable[[int, int], torch.Tensor],
                  encode_score: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
                  device, rank: int = 0, world: int = 1, gather: bool = True, sync: Optional[Callable[[], None]] = None,
                  progress: Optional[Callable[[int, int], None]] = None, gather_dst: Optional[int] = None,
                  gather_device=None) -> Dict:
    """Runs this rank's shard of the job.  `encode_score(crops) -> (emb [b, crops, E], score [b, score_dim])` on `device`.
    Returns {'emb_local', 'score_local'} (this rank's block) and {'emb', 'score'} (gather=True: the FULL job on every rank, or -- with gather_dst=r -- on rank r only and None
    elsewhere; gather=False: the local block), 'n_local', 'lo', 'hi', 'batches' and wall-clock seconds of the encode phase
    and of the gather ('t_encode', 't_gather'; `sync` is called before each clock is read -- torch.cuda.synchronize on a
    GPU).  The gather allocates nothing but the result (sharding.gather_rows).  `gather_device` (rehearsals only: several ranks
    on one GPU under gloo) stages the local blocks there before the exchange; RCCL gathers the device tensors in place."""
    if n_images < 0 or batch_images < 1:
        raise ValueError(f"bad job shape: {n_images} images in batches of {batch_images}")
    lo, hi = shard_bounds(n_images, rank, world)
    n_local = hi - lo
    emb = torch.empty((n_local, crops_per_image, embed_dim), dtype=torch.float32, device=device)
    score = torch.empty((n_local, score_dim), dtype=torch.float32, device=device)
    sync = sync or (lambda: None)
    sync()
    t0 = time.perf_counter()
    batches = 0
    for b0 in range(0, n_local, batch_images):
        nb = min(batch_images, n_local - b0)                  # ragged last batch: encoded at its own size
        crops = source(lo + b0, nb)
        e, s = encode_score(crops)
        emb[b0:b0 + nb].copy_(e.view(nb, crops_per_image, embed_dim))
        score[b0:b0 + nb].copy_(s.view(nb, score_dim))
        batches += 1
        if progress is not None:
            progress(b0 + nb, n_local)
    sync()
    t1 = time.perf_counter()
    out = {"n_local": n_local, "lo": lo, "hi": hi, "batches": batches, "t_encode": t1 - t0, "t_gather": 0.0,
           "emb_local": emb, "score_local": score}            # this rank's own rows, whatever the gather does
    in_group = dist.is_available() and dist.is_initialized()
    if gather and (world > 1 or in_group):                    # (a group of ONE rank gathers through the backend too)
        if not in_group:
            raise RuntimeError("world > 1 needs an initialised torch.distributed process group")
        ge, gs = (emb, score) if gather_device is None else (emb.to(gather_device), score.to(gather_device))
        full_e = gather_rows(ge, n_images, dst=gather_dst)    # the one exchange of the path
        full_s = gather_rows(gs, n_images, dst=gather_dst)
        sync()
        out["t_gather"] = time.perf_counter() - t1
        out["emb"], out["score"] = full_e, full_s
    else:
        out["emb"], out["score"] = emb, score
    return out
