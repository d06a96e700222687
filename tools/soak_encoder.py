#!/usr/bin/env python3
"""Developer race screen: the full-size encoder (bf16 and fp8 block GEMMs) run back to back many times on the same input
must be bitwise identical every time -- hand-placed waits show ordering bugs as run-to-run differences."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
dev = torch.device("cuda", 0)
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ARCH = sys.argv[2] if len(sys.argv) > 2 else "ViT-L-14"        # ViT-L-14-336: the long streaming attention kernel; ViT-H-14: head dim 80
cfg = vit_config.config_for(ARCH + "/laion2b") if ARCH in ("ViT-H-14", "ViT-g-14", "ViT-bigG-14") else vit_config.ARCHS[ARCH]   # (g / bigG: the zero-padded towers)
sd = vit_config.seeded_state_dict(cfg, 0)
big = 2048 if cfg.tokens < 300 else 640
for prec, crops_n in (("bf16", big), ("fp8", big), ("bf16", 333), ("fp8", 61)):     # full tiles and ragged last tiles
    vit = HipViT(cfg, sd, dev, precision=prec)
    g = torch.Generator(device=dev).manual_seed(crops_n)
    crops = torch.randn(crops_n, 3, cfg.image_size, cfg.image_size, device=dev, generator=g)
    first = vit.encode(crops).clone()
    assert torch.isfinite(first).all()
    bad = 0
    t0 = time.time()
    for i in range(n_rep):
        out = vit.encode(crops)
        if not torch.equal(out, first):
            bad += 1
            print(f"  {prec} {crops_n} crops: run {i} differs, max |d| = {(out - first).abs().max().item():.3e}", flush=True)
    print(f"{prec} {crops_n} crops: {n_rep} repeats, {bad} differing, {time.time() - t0:.1f} s", flush=True)
    vit.close(); del vit
