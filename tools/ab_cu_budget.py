#!/usr/bin/env python3
"""Developer: what a CU budget costs the encoder alone (clipenc_set_cu_budget: the persistent kernels' grids shrink to n CUs and the
rest of the chip idles): ViT-L/14, 512 images x 4 crops resident in HBM, bf16 and fp8, budgets interleaved over three rounds on one box.
On a power-capped board the CUs taken away return most of their share as clock (tools/experiments/README.md, round 5)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
dev = torch.device("cuda", 0)
cfg = vit_config.ARCHS["ViT-L-14"]
vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev)
g = torch.Generator(device=dev).manual_seed(1234)
crops = torch.randint(0, 256, (2048, 3, 224, 224), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
budgets = [int(b) for b in os.environ.get("BUDGETS", "0,248,240,224,192").split(",")]
ref = {}
for prec in ("bf16", "fp8"):
    vit.set_precision(prec)
    for rnd in range(3):
        for b in budgets:
            vit.set_cu_budget(b)
            e = vit.encode(crops); torch.cuda.synchronize()
            if prec not in ref: ref[prec] = e.clone()
            assert torch.equal(e, ref[prec]), f"budget {b} changed the {prec} embeddings"
            t0 = time.perf_counter()
            for _ in range(4): vit.encode(crops)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 4
            print(f"{prec} round {rnd} budget {b:3d}: {dt * 1e3:7.2f} ms  {512 / dt:7.1f} images/s", flush=True)
vit.set_cu_budget(0)
