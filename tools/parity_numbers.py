#!/usr/bin/env python3
"""Developer: the measured distance of the GPU embeddings (bf16 and fp8 block GEMMs) to the fp32 oracle on seeded ViT-L/14
(6 crops), the numbers DESIGN.md quotes next to the 1e-3 tolerance of the tests."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from oracle import vit_oracle
from tests.helpers import one_minus_cos, synthetic_crops
cfg = vit_config.ARCHS["ViT-L-14"]; sd = vit_config.seeded_state_dict(cfg, 0); dev = torch.device("cuda", 0)
crops = synthetic_crops(6, cfg.image_size, int(os.environ.get("PARITY_SEED", "77")))
torch.set_num_threads(32)
ref = vit_oracle.encode_image(sd, cfg, crops)
for prec in ("bf16", "fp8"):
    vit = HipViT(cfg, sd, dev, precision=prec)
    got = vit.encode(crops.to(dev)).cpu()
    print(prec, "1-cos max %.3e  max abs %.3e" % (one_minus_cos(got, ref).max().item(), (got - ref).abs().max().item()))
    vit.close()
