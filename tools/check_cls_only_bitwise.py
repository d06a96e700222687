#!/usr/bin/env python3
"""Developer check: the last block on the class-token rows only (default) gives the same embedding BITS as running the last
block on every token (CLIPENC_FULL_LAST_BLOCK=1), bf16 and fp8, ViT-L/14, 64 crops.  Runs the second configuration in a child
process (the switch is read at clipenc_create)."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT


def run(path):
    dev = torch.device("cuda", 0)
    cfg = vit_config.ARCHS["ViT-L-14"]
    vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev)
    g = torch.Generator(device=dev).manual_seed(3)
    crops = torch.randn(64, 3, 224, 224, device=dev, generator=g)
    out = {"bf16": vit.encode(crops).cpu()}
    vit.set_precision("fp8")
    out["fp8"] = vit.encode(crops).cpu()
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    a, b = "/tmp/cls_only.pt", "/tmp/full_last.pt"
    subprocess.check_call([sys.executable, __file__, a])
    subprocess.check_call([sys.executable, __file__, b], env=dict(os.environ, CLIPENC_FULL_LAST_BLOCK="1"))
    x, y = torch.load(a), torch.load(b)
    for k in ("bf16", "fp8"):
        same = torch.equal(x[k], y[k])
        d = (x[k] - y[k]).abs().max().item()
        print(f"{k}: class-token-only last block vs full last block: bitwise equal = {same}, max abs diff = {d:.3e}")
