#!/usr/bin/env python3
"""Developer: the measured distance of the GPU embeddings (bf16 and fp8 block GEMMs) to the fp32 oracle, the numbers DESIGN.md
quotes next to the 1e-3 tolerance of the tests.

    python tools/parity_numbers.py                                  seeded ViT-L/14 weights (what the tests run)
    python tools/parity_numbers.py --model ViT-L-14-336/openai --model_path W [--images DIR]
        REAL weights from a local file / directory (vit_config.load_weights: OpenAI TorchScript archive, open_clip or
        transformers checkpoint): what a user with pretrained weights runs to check the bf16 and the e4m3 tower against the fp32
        CPU path on their own activation statistics (real checkpoints carry outlier channels that seeded weights do not).
        --images: a directory of image files, centre crops through the reference's val transform; default: synthetic pixels.
The oracle runs on the host (about 10 s per crop at ViT-L/14 on 32 threads): --crops bounds it."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clip_assisted_data_labeling_amd import vit_config  # noqa: E402
from clip_assisted_data_labeling_amd.embedder import HipViT  # noqa: E402
from oracle import vit_oracle  # noqa: E402
from tests.helpers import one_minus_cos, synthetic_crops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L-14/openai", help="'<arch>/<pretrained>' as in _1_embed_with_CLIP.py")
    ap.add_argument("--model_path", default=None, help="local weights (file or directory); default: seeded weights")
    ap.add_argument("--images", default=None, help="directory of images (centre crop each); default: synthetic uint8 pixels")
    ap.add_argument("--crops", type=int, default=6)
    ap.add_argument("--seed", type=int, default=int(os.environ.get("PARITY_SEED", "77")))
    args = ap.parse_args()
    cfg = vit_config.config_for(args.model)
    if args.model_path:
        sd = vit_config.normalise_state_dict(vit_config.load_weights(args.model, args.model_path), cfg)
        sd = {k: v.float() for k, v in sd.items()}
    else:
        sd = vit_config.seeded_state_dict(cfg, 0)
    if args.images:
        from PIL import Image
        from clip_assisted_data_labeling_amd.preprocess import clip_val_transform
        tf = clip_val_transform(cfg.image_size)
        names = sorted(f for f in os.listdir(args.images) if f.lower().endswith((".jpg", ".jpeg", ".png", ".webp")))[:args.crops]
        crops = torch.stack([tf(Image.open(os.path.join(args.images, f)).convert("RGB")) for f in names])
    else:
        crops = synthetic_crops(args.crops, cfg.image_size, args.seed)
    dev = torch.device("cuda", 0)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = vit_oracle.encode_image(sd, cfg, crops)
    print(f"{args.model} ({'weights from ' + args.model_path if args.model_path else 'seeded weights'}), {crops.shape[0]} crops; "
          "north_star tolerance: 1 - cos < 1e-3")
    for prec in ("bf16", "fp8"):
        vit = HipViT(cfg, sd, dev, precision=prec)
        got = vit.encode(crops.to(dev)).cpu()
        print(prec, "1-cos max %.3e  max abs %.3e" % (one_minus_cos(got, ref).max().item(), (got - ref).abs().max().item()))
        vit.close()


if __name__ == "__main__":
    main()
