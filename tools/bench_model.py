#!/usr/bin/env python3
"""Developer timing of clipenc_encode for any architecture of vit_config.ARCHS (e.g. ViT-L-14-336, ViT-B-32)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from tools.quick_bench import timeit

ap = argparse.ArgumentParser(); ap.add_argument("--arch", default="ViT-L-14-336"); ap.add_argument("--crops", type=int, default=512)
args = ap.parse_args()
cfg = vit_config.ARCHS[args.arch]
dev = torch.device("cuda", 0)
vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev)
crops = torch.randn(args.crops, 3, cfg.image_size, cfg.image_size, device=dev)
vit.profile_enable(True)
ms = timeit(lambda: vit.encode(crops), iters=3, warmup=1)
prof = vit.profile_read()
fl = 2.0 * cfg.macs_per_crop() * args.crops
print(f"{args.arch}: {args.crops} crops in {ms:.1f} ms = {args.crops / ms * 1e3:.0f} crops/s = {args.crops / 4 / ms * 1e3:.0f} img/s, "
      f"{fl / ms / 1e9:.0f} TFLOP/s ({fl / ms / 1e9 / 2516.6 * 100:.1f}% of peak)")
print({k: round(v[0] / 4, 2) for k, v in prof.items() if v[0] > 0 and not k.startswith('shape:')})
