#!/usr/bin/env python3
"""Developer: |mean| / std of the residual-stream rows of the seeded ViT-L/14 at several depths -- the quantity that scales the e4m3 noise
of the fused fp8 tower (a row with mean t sigma carries sqrt(1 + t^2) times the noise of a centred one)."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from tests.helpers import synthetic_crops
dev = torch.device("cuda", 0)
cfg = vit_config.ARCHS["ViT-L-14"]; sd = vit_config.seeded_state_dict(cfg, 0)
vit = HipViT(cfg, sd, dev)
crops = synthetic_crops(8, cfg.image_size, 77).to(dev)
for nl in (0, 1, 6, 12, 23, 24):
    x = vit.forward_tokens(crops, nl).float()
    t = (x.mean(-1) / x.std(-1)).abs()
    print(f"after {nl:2d} blocks: |mean|/std of the residual rows: median {t.median().item():.3f}  99% {t.flatten().quantile(0.99).item():.3f}  max {t.max().item():.3f}")
