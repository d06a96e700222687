import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from oracle import vit_oracle
from tests.helpers import one_minus_cos, synthetic_crops
dev = torch.device("cuda", 0)
for name, n in (("ViT-H-tiny-test/laion2b", 9), ("ViT-H-mid-test/seed0", 6), ("ViT-H-14/laion2b_s32b_b79k", 3)):
    cfg = vit_config.config_for(name); sd = vit_config.seeded_state_dict(cfg, 3)
    crops = synthetic_crops(n, cfg.image_size, 5)
    ref = vit_oracle.encode_image(sd, cfg, crops)
    vit = HipViT(cfg, sd, dev)
    b = vit.encode(crops.to(dev)).cpu()
    vit.set_precision("fp8")
    f = vit.encode(crops.to(dev)).cpu()
    f2 = vit.encode(crops.to(dev)).cpu()
    print(name, "bf16", one_minus_cos(b, ref).max().item(), "fp8", one_minus_cos(f, ref).max().item(), "repeat equal", bool(torch.equal(f, f2)), "finite", bool(torch.isfinite(f).all()))
    if "H-14/" in name:
        import time
        big = torch.randn(2048, 3, 224, 224, device=dev)
        for prec in ("bf16", "fp8"):
            vit.set_precision(prec); vit.encode(big); torch.cuda.synchronize(); t0 = time.perf_counter(); vit.encode(big); vit.encode(big); torch.cuda.synchronize()
            print(prec, "2048 crops:", (time.perf_counter() - t0) / 2 * 1e3, "ms ->", 512 / ((time.perf_counter() - t0) / 2), "img/s")
    vit.close()
