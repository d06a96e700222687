"""The last transformer block runs on the class-token rows only: the embedding is ln_post + proj of
token 0 (/root/reference/utils/embedder.py:98 takes the pooled output), so the rows left out are dead.  Proven here bit for
bit: the diagnostic library (same sources, -DCLIPENC_DIAG) reads CLIPENC_FULL_LAST_BLOCK=1 at clipenc_create and then runs the
last block on every token; ViT-L/14, 64 crops, bf16 and fp8, each configuration in a child process of its own."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_LIB = os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so")

pytestmark = pytest.mark.gpu

_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
dev = torch.device("cuda", 0)
cfg = vit_config.ARCHS["ViT-L-14"]
sd = vit_config.seeded_state_dict(cfg, 0)
if len(sys.argv) > 3 and sys.argv[3] == "planted":
    # residual channels 100-300x the median and rows with non-zero mean at every depth (tests/test_gpu_configs.py::_plant_outliers):
    # what real checkpoints look like and seeded weights do not -- |z . W'_v| and |x . r_h| are then far larger than |v| and |q . k|
    gg = torch.Generator().manual_seed(0)
    hot = torch.randperm(cfg.width, generator=gg)[:5]
    sd["ln_pre.weight"][hot] *= torch.tensor([50.0, 70.0, 90.0, 110.0, 40.0])
    sd["ln_pre.bias"] += 1.5
    sd["ln_pre.bias"][hot] += torch.tensor([60.0, -80.0, 100.0, -40.0, 50.0])
    for l in range(cfg.layers):
        sd[f"transformer.resblocks.{l}.attn.out_proj.bias"][hot[l % 5]] += 4.0
        sd[f"transformer.resblocks.{l}.mlp.c_proj.bias"][hot[(l + 2) % 5]] -= 4.0
vit = HipViT(cfg, sd, dev)
g = torch.Generator(device=dev).manual_seed(3)
crops = torch.randn(64, 3, 224, 224, device=dev, generator=g)
out = {"bf16": vit.encode(crops).cpu()}
vit.set_precision("fp8")
out["fp8"] = vit.encode(crops).cpu()
torch.save(out, sys.argv[1])
"""


def _run(path, env_extra, weights="seeded"):
    env = {k: v for k, v in os.environ.items() if k not in ("CLIPENC_FULL_LAST_BLOCK", "CLIPENC_CLS_KV", "CLIPENC_LIB_PATH")}
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", _CHILD, path, ROOT, weights], env=env, check=True, timeout=600)
    return torch.load(path)


def test_cls_only_last_block_equals_full_last_block_bitwise(gpu, tmp_path):
    """Three forms of the last block: (a) every token (CLIPENC_FULL_LAST_BLOCK=1), (b) class-token rows only with K | V projected for
    every token (CLIPENC_CLS_KV=1: rounds 2-3), (c) the shipped one: class-token rows only and the attention WITHOUT K and V
    (cls_attention.hip: the LayerNorm fold lets the one query per head be scored against the raw residual rows).  (a) == (b) bit for
    bit -- the rows left out are dead; (c) is the same mathematics in another order of bf16 roundings (no rounded K / V, weights
    applied to x), held to (b) far inside the embedding tolerance."""
    assert os.path.exists(DIAG_LIB), f"{DIAG_LIB} missing: __graft_entry__.build() makes it (make -C .../csrc diag)"
    product = _run(str(tmp_path / "product.pt"), {})                                          # the shipped library, default path: (c)
    short = _run(str(tmp_path / "short.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB})                  # diagnostic build, same path
    cls_only = _run(str(tmp_path / "cls.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB, "CLIPENC_CLS_KV": "1"})                                  # (b)
    full = _run(str(tmp_path / "full.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB, "CLIPENC_CLS_KV": "1", "CLIPENC_FULL_LAST_BLOCK": "1"})     # (a)
    for prec, tol in (("bf16", 2e-5), ("fp8", 1e-4)):
        assert torch.isfinite(product[prec]).all()
        assert torch.equal(product[prec], short[prec]), f"{prec}: diagnostic build differs from the product library"
        assert torch.equal(cls_only[prec], full[prec]), \
            f"{prec}: class-token-only last block differs from the full last block, max abs {(cls_only[prec] - full[prec]).abs().max():.3e}"
        assert not torch.equal(product[prec], cls_only[prec])                                  # (c) did run
        omc = 1.0 - (product[prec].double() * cls_only[prec].double()).sum(-1)
        print(f"{prec}: K/V-free class-token attention vs the projected one: max 1 - cos {omc.max().item():.2e}")
        assert omc.max().item() < tol, (prec, omc.max().item())
    # the switch must not be live in the PRODUCT library: with the variable set it still takes the class-token-only path
    # (same bits either way, so this is checked through the library's symbol table instead)
    strings = subprocess.run(["strings", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip.so")],
                             capture_output=True, text=True).stdout
    for name in ("CLIPENC_FULL_LAST_BLOCK", "CLIPENC_CLS_KV", "CLIPENC_ATTN_IMPL", "CLIPENC_ATTN_DBG"):
        assert name not in strings, f"developer switch {name} is compiled into the product library"


def test_kv_free_class_token_attention_on_a_tower_with_outlier_channels(gpu, tmp_path):
    """(b) against (c) again on the planted-outlier tower: token means far from zero and channels hundreds of times the median make
    |z . W'_v| much larger than |v|; o = Of - m colsum + bias is therefore formed in fp32 and rounded once (cls_finish_kernel)."""
    short = _run(str(tmp_path / "short.pt"), {}, "planted")                                                          # (c), product library
    cls_only = _run(str(tmp_path / "cls.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB, "CLIPENC_CLS_KV": "1"}, "planted")      # (b)
    for prec, tol in (("bf16", 5e-5), ("fp8", 2e-4)):
        assert torch.isfinite(short[prec]).all() and not torch.equal(short[prec], cls_only[prec])
        omc = 1.0 - (short[prec].double() * cls_only[prec].double()).sum(-1)
        print(f"{prec}, planted outliers: K/V-free class-token attention vs the projected one: max 1 - cos {omc.max().item():.2e}")
        assert omc.max().item() < tol, (prec, omc.max().item())


_CHILD_TAIL = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
dev = torch.device("cuda", 0)
cfg = vit_config.ARCHS["ViT-L-14"]
vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev)
g = torch.Generator(device=dev).manual_seed(5)
# 700 crops = 179 900 token rows = 703 row tiles: 2 812 / 8 436 / 11 248 tiles = 10.98 / 32.95 / 43.9 rounds of 256 workgroups
crops = torch.randint(0, 256, (700, 3, 224, 224), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
out = {"a": vit.encode(crops).cpu(), "b": vit.encode(crops).cpu(), "tokens": vit.forward_tokens(crops[:300]).float().cpu()}
vit.set_precision("fp8")
out["fp8"] = vit.encode(crops).cpu()
out["fp8_b"] = vit.encode(crops).cpu()
torch.save(out, sys.argv[1])
"""


def test_tiles_handed_out_by_ticket_give_the_bits_of_the_strided_walk(gpu, tmp_path):
    """The persistent GEMMs hand the tiles of a launch's last one-to-two rounds out from a ticket counter, in the order the
    workgroups get there (gemm_persist.hip): WHICH workgroup runs a tile must not change a bit.  The diagnostic library reads
    CLIPENC_STATIC_TILES=1 at clipenc_create and then walks every tile by stride, as rounds 1-3 did."""
    def run(path, env_extra):
        env = {k: v for k, v in os.environ.items() if k not in ("CLIPENC_STATIC_TILES", "CLIPENC_LIB_PATH")}
        env.update(env_extra)
        subprocess.run([sys.executable, "-c", _CHILD_TAIL, path, ROOT], env=env, check=True, timeout=600)
        return torch.load(path)
    product = run(str(tmp_path / "p.pt"), {})
    ticket = run(str(tmp_path / "t.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB})
    strided = run(str(tmp_path / "s.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB, "CLIPENC_STATIC_TILES": "1"})
    assert torch.isfinite(product["a"]).all() and torch.equal(product["a"], product["b"])          # run to run, whatever the hand-out order was
    assert torch.equal(product["fp8"], product["fp8_b"]) and not torch.equal(product["fp8"], product["a"])
    for k in ("a", "b", "tokens", "fp8", "fp8_b"):
        assert torch.equal(product[k], ticket[k]) and torch.equal(ticket[k], strided[k]), k
    strings = subprocess.run(["strings", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip.so")],
                             capture_output=True, text=True).stdout
    assert "CLIPENC_STATIC_TILES" not in strings


_CHILD_K128 = r"""
import sys, dataclasses, torch
sys.path.insert(0, sys.argv[2])
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
dev = torch.device("cuda", 0)
# patch 4 -> 48 values per patch -> K padded to 128: the patch GEMM's tile is ONE stage pair; 4 200 crops x 64 patches = 1 050 tiles
cfg = dataclasses.replace(vit_config.ARCHS["ViT-tiny-test"], image_size=32, patch=4, width=256, layers=1, heads=4, mlp_dim=512, embed_dim=32)
vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 2), dev)
g = torch.Generator(device=dev).manual_seed(9)
crops = torch.randn(4200, 3, 32, 32, device=dev, generator=g)
torch.save({"a": vit.encode(crops).cpu(), "b": vit.encode(crops).cpu(), "crops": crops[:4].cpu()}, sys.argv[1])
"""


def test_patch_gemm_of_one_stage_pair_walks_by_stride(gpu, tmp_path):
    """K = 128 (a patch GEMM with kpad = 128): the tile is its last stage pair, so no stage barrier lies between the ticket
    slot's write and its read -- such launches take no tickets (gemm_persist.hip).  Over four rounds of tiles the product must
    equal the strided walk of the diagnostic library bit for bit, run to run, and match the oracle."""
    import dataclasses
    from clip_assisted_data_labeling_amd import vit_config
    from oracle import vit_oracle

    def run(path, env_extra):
        env = {k: v for k, v in os.environ.items() if k not in ("CLIPENC_STATIC_TILES", "CLIPENC_LIB_PATH")}
        env.update(env_extra)
        subprocess.run([sys.executable, "-c", _CHILD_K128, path, ROOT], env=env, check=True, timeout=600)
        return torch.load(path)
    product = run(str(tmp_path / "p.pt"), {})
    strided = run(str(tmp_path / "s.pt"), {"CLIPENC_LIB_PATH": DIAG_LIB, "CLIPENC_STATIC_TILES": "1"})
    assert torch.isfinite(product["a"]).all() and torch.equal(product["a"], product["b"])
    assert torch.equal(product["a"], strided["a"]) and torch.equal(strided["a"], strided["b"])
    cfg = dataclasses.replace(vit_config.ARCHS["ViT-tiny-test"], image_size=32, patch=4, width=256, layers=1, heads=4, mlp_dim=512, embed_dim=32)
    ref = vit_oracle.encode_image(vit_config.seeded_state_dict(cfg, 2), cfg, product["crops"])
    omc = 1.0 - (product["a"][:4].double() * ref.double()).sum(-1)
    assert omc.max().item() < 1e-3, omc
