#!/usr/bin/env python3
"""Developer: do two builds of the library produce the same embedding BITS?  Each build runs in a child process
(CLIPENC_LIB_PATH); ViT-L/14 seeded weights, 64 fp32 crops + the same crops as uint8, bf16 and fp8, plus ViT-B-32 (patch 32).
    python tools/ab_bits.py <suffix|cur> <suffix|cur>      suffix -> clip_assisted_data_labeling_amd/libclipenc_hip_<suffix>.so"""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(path):
    from clip_assisted_data_labeling_amd import vit_config
    from clip_assisted_data_labeling_amd.embedder import HipViT
    dev = torch.device("cuda", 0)
    out = {}
    for arch, n in (("ViT-L-14", 64), ("ViT-B-32", 40)):
        cfg = vit_config.ARCHS[arch]
        vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev)
        g = torch.Generator(device=dev).manual_seed(3)
        crops = torch.randn(n, 3, cfg.image_size, cfg.image_size, device=dev, generator=g)
        u8 = torch.randint(0, 256, (n, 3, cfg.image_size, cfg.image_size), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
        out[arch + " bf16 f32-in"] = vit.encode(crops).cpu()
        out[arch + " bf16 f16-in"] = vit.encode(crops.half()).cpu()
        out[arch + " bf16 u8-in"] = vit.encode(u8).cpu()
        out[arch + " tokens"] = vit.forward_tokens(crops[:8]).float().cpu()
        vit.set_precision("fp8")
        out[arch + " fp8 f32-in"] = vit.encode(crops).cpu()
        vit.close()
    torch.save(out, path)


def lib(sfx):
    return os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip.so" if sfx == "cur" else f"libclipenc_hip_{sfx}.so")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        run(sys.argv[2])
        sys.exit(0)
    a, b = sys.argv[1], sys.argv[2]
    outs = []
    for sfx in (a, b):
        path = f"/tmp/ab_bits_{sfx}.pt"
        subprocess.check_call([sys.executable, __file__, "--child", path], env=dict(os.environ, CLIPENC_LIB_PATH=lib(sfx)))
        outs.append(torch.load(path))
    ok = True
    for k in outs[0]:
        same = torch.equal(outs[0][k], outs[1][k])
        ok &= same
        print(f"{k:28s} {a} vs {b}: bitwise equal = {same}, max abs diff = {(outs[0][k] - outs[1][k]).abs().max().item():.3e}")
    sys.exit(0 if ok else 1)
