"""ViT-H-14 shapes: width 1280 = 16 heads of 80, a 5 120-wide erf-GELU MLP, five 256-column row-statistics parts (open_clip's
"ViT-H-14/<laion tag>", which /root/reference/utils/embedder.py:63-73 can name like any "<arch>/<tag>").  What differs from ViT-L on the
device: the head-dim-80 attention kernel (attn_hd_kernel: five k steps, O^T in 2.5 tiles of d), the row statistics of the
LayerNorm-folded GEMMs added up over five parts by a pass of their own (the GEMM's LDS layout holds four), the last block's
class-token attention on projected K | V (the K-and-V-free shortcut is head-dim-64 code), and every GEMM's tile counts.
Tolerance: north_star -- 1 - cos < 1e-3 against the fp32 CPU oracle."""
import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib, vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from oracle import vit_oracle
from tests.helpers import one_minus_cos, synthetic_crops

pytestmark = pytest.mark.gpu
COS_TOL = 1e-3


def _stream(dev):
    return _lib.current_stream_ptr(dev)


@pytest.mark.parametrize("n_crops,n_tok,heads", [(2, 5, 16), (3, 50, 16), (2, 257, 16), (1, 288, 16), (5, 33, 4), (2, 272, 4), (3, 1, 16),
                                                  (2, 64, 16), (1, 273, 16)])
def test_attention_head_dim_80_matches_fp32_reference(gpu, n_crops, n_tok, heads):
    lib = _lib.load()
    width = heads * 80
    g = torch.Generator().manual_seed(n_tok + heads)
    qkv = (torch.randn(n_crops * n_tok, 3 * width, generator=g) * 1.5).to(torch.bfloat16)
    out = torch.full((n_crops * n_tok, width), float("nan"), dtype=torch.bfloat16, device=gpu)
    qkv_dev = qkv.to(gpu)
    _lib.check(lib.clipenc_op_attention(qkv_dev.data_ptr(), out.data_ptr(), n_crops, n_tok, width, heads, _stream(gpu)), "attention")
    torch.cuda.synchronize()
    q, k, v = qkv.float().view(n_crops, n_tok, 3, heads, 80).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(q @ k.transpose(-1, -2) * 80.0 ** -0.5, -1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(n_crops * n_tok, width)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 0.03       # bf16 weights and bf16 output rounding on |v| ~ 1.5
    assert one_minus_cos(got, ref).max().item() < 2e-4
    again = torch.empty_like(out)
    _lib.check(lib.clipenc_op_attention(qkv_dev.data_ptr(), again.data_ptr(), n_crops, n_tok, width, heads, _stream(gpu)), "attention")
    assert torch.equal(out, again)


def test_attention_head_dim_80_one_dominant_key_and_asymmetric_values(gpu):
    """A key that towers over every row (the true-max subtraction) and values that identify their own (key, column): a transposed or
    permuted V read, or a wrong d tile, cannot pass."""
    lib = _lib.load()
    n_tok, heads = 100, 16
    width = heads * 80
    qkv = torch.zeros(n_tok, 3 * width)
    qkv[:, :width] = 20.0
    qkv[37, width:2 * width] = 20.0                                 # key 37: logit 20 * 20 * 80 / sqrt(80)
    qkv[:, 2 * width:] = (torch.arange(n_tok).view(-1, 1) * 0.25 + torch.arange(width).view(1, -1) / 256.0)
    out = torch.empty((n_tok, width), dtype=torch.bfloat16, device=gpu)
    _lib.check(lib.clipenc_op_attention(qkv.to(torch.bfloat16).to(gpu).data_ptr(), out.data_ptr(), 1, n_tok, width, heads, _stream(gpu)), "attention")
    want = qkv[37, 2 * width:].to(torch.bfloat16).float().view(1, width).expand(n_tok, width)
    assert torch.allclose(out.float().cpu(), want, atol=2e-2, rtol=0)


@pytest.mark.parametrize("arch,tag,n_crops", [("ViT-H-tiny-test", "seed0", 7), ("ViT-H-tiny-test", "laion2b", 300), ("ViT-H-mid-test", "laion2b", 5),
                                             ("ViT-H-mid-test", "seed0", 70)])
def test_vit_h_shapes_match_fp32_oracle(gpu, arch, tag, n_crops):
    cfg = vit_config.config_for(f"{arch}/{tag}")
    assert cfg.width == 1280 and cfg.width // cfg.heads == 80 and cfg.mlp_dim == 5120
    sd = vit_config.seeded_state_dict(cfg, 3)
    crops = synthetic_crops(n_crops, cfg.image_size, 80 + n_crops)
    ref = vit_oracle.encode_image(sd, cfg, crops[:8])              # the oracle on the first crops; the rest must agree with each other
    vit = HipViT(cfg, sd, gpu)
    try:
        emb = vit.encode(crops.to(gpu))
        assert emb.shape == (n_crops, cfg.embed_dim) and torch.isfinite(emb).all()
        assert torch.equal(emb, vit.encode(crops.to(gpu)))          # bitwise repeatable
        omc = one_minus_cos(emb[:8].cpu(), ref)
        print(f"{arch}/{tag} 1-cos vs fp32 oracle:", omc)
        assert omc.max().item() < COS_TOL, omc
        # a crop's embedding does not depend on its batch (row independence of every kernel, ragged last tiles)
        alone = vit.encode(crops[:3].to(gpu))
        assert torch.equal(alone, emb[:3])
        if cfg.tokens >= 32:
            # the e4m3 block GEMMs (the unfused tower: row-quantised operands, static scales for O and the hidden rows); the 5-token tower's
            # statistics are too thin for the 1e-3 budget (1.0e-3 measured), the 50-token one and the full size are held to it
            vit.set_precision("fp8")
            e8 = vit.encode(crops.to(gpu))
            omc8 = one_minus_cos(e8[:8].cpu(), ref)
            print(f"{arch}/{tag} fp8 1-cos vs fp32 oracle:", omc8)
            assert torch.isfinite(e8).all() and omc8.max().item() < COS_TOL, omc8
            assert torch.equal(e8, vit.encode(crops.to(gpu)))
    finally:
        vit.close()


def test_vit_h_14_full_size_matches_the_independent_implementation(gpu, golden_dir):
    """tests/golden/encoder_ViT-H-14-erf.npz (`make_golden.py vit_h`): the embeddings of transformers.CLIPVisionModelWithProjection
    (hidden_act = gelu) on the seeded FULL-SIZE tower -- 1280 wide x 32 blocks x 257 tokens, 632 M parameters -- with the oracle asserted
    within 1e-5 of them in the authoring container.  The HIP tower is held to the north_star tolerance against THOSE vectors; the taps pin
    the front end (ln_pre), the first block and token 1 behind the last block separately."""
    import os
    g = np.load(os.path.join(golden_dir, "encoder_ViT-H-14-erf.npz"))
    cfg = vit_config.config_for(f"{str(g['arch'])}/{str(g['pretrained'])}")
    assert (cfg.width, cfg.layers, cfg.tokens) == (1280, 32, 257) and cfg.act == vit_config.ACT_GELU_ERF
    sd = vit_config.seeded_state_dict(cfg, int(g["weight_seed"]))
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_abs_sum"])) <= 1e-9 * wsum, "the seeded weights are not the ones the fixture was made with"
    crops = synthetic_crops(int(g["n_crops"]), cfg.image_size, int(g["input_seed"]))
    assert abs(float(crops.double().abs().sum()) - float(g["crops_abs_sum"])) <= 1e-9 * float(g["crops_abs_sum"])
    hf = torch.from_numpy(g["emb_transformers"])
    vit = HipViT(cfg, sd, gpu)
    try:
        got = vit.encode(crops.to(gpu)).cpu()
        omc = one_minus_cos(got, hf)
        print("ViT-H-14 bf16 1-cos vs transformers:", omc.max().item())
        assert omc.max().item() < COS_TOL, omc
        assert (got - hf).abs().max().item() < 0.02
        x0 = vit.debug_run_layers(crops.to(gpu), 0).float().cpu()
        assert one_minus_cos(x0[:, 0], torch.from_numpy(g["ln_pre_cls"])).max().item() < 1e-4
        x1 = vit.debug_run_layers(crops.to(gpu), 1).float().cpu()
        assert one_minus_cos(x1[:, 0], torch.from_numpy(g["block0_cls"])).max().item() < 1e-4
        xl = vit.debug_run_layers(crops.to(gpu), cfg.layers).float().cpu()
        assert one_minus_cos(xl[:, 1], torch.from_numpy(g["last_block_tok1"])).max().item() < 5e-4
        vit.set_precision("fp8")                                     # e4m3 block GEMMs at full size, against the same vectors
        got8 = vit.encode(crops.to(gpu)).cpu()
        omc8 = one_minus_cos(got8, hf)
        print("ViT-H-14 fp8 1-cos vs transformers:", omc8.max().item())
        assert omc8.max().item() < COS_TOL, omc8
        vit.set_precision("bf16")
        # properties at a batch that fills the chip's tiles raggedly (200 crops = 51 400 token rows = 200.8 row tiles)
        gen = torch.Generator(device=gpu).manual_seed(9)
        big = torch.randn(200, 3, 224, 224, device=gpu, generator=gen)
        big[:2] = crops.to(gpu)
        e = vit.encode(big)
        assert torch.isfinite(e).all() and torch.allclose(e.norm(dim=-1), torch.ones(200, device=gpu), atol=1e-5)
        assert torch.equal(e[:2].cpu(), got)                         # a crop's embedding does not depend on its batch
        assert torch.equal(e, vit.encode(big))
    finally:
        vit.close()
