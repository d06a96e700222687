"""ViT-L-14-336 at FULL size (1024 wide x 24 blocks x 577 tokens): the reference's DEFAULT model
(/root/reference/_1_embed_with_CLIP.py:190) and the tower its only shipped checkpoint was trained on
(`clip_models=['ViT-L-14-336/openai']`, so `AestheticRegressor`, /root/reference/utils/embedder.py:277-311, runs THIS tower).
577 tokens take the single-pass long attention kernel (attn_long_kernel) and a different tile count in every GEMM, so the
2-layer `ViT-long-test` of test_gpu_parity.py does not stand in for it.

Tolerances: north_star -- embeddings 1 - cos < 1e-3 against the fp32 CPU oracle, scores within 1e-4 abs."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from clip_assisted_data_labeling_amd.nn_model import HipRegressor, SimpleFC
from oracle import fcreg_oracle, vit_oracle
from tests.helpers import one_minus_cos, synthetic_crops

pytestmark = pytest.mark.gpu

COS_TOL, SCORE_TOL = 1e-3, 1e-4
ARCH = "ViT-L-14-336"


@pytest.fixture(scope="module")
def tower(gpu):
    cfg = vit_config.ARCHS[ARCH]
    assert (cfg.tokens, cfg.width, cfg.layers) == (577, 1024, 24)
    sd = vit_config.seeded_state_dict(cfg, 0)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    crops = synthetic_crops(6, cfg.image_size, 336)
    taps = {}
    ref = vit_oracle.encode_image(sd, cfg, crops, taps)                      # fp32 CPU oracle, 6 crops
    vit = HipViT(cfg, sd, gpu)
    yield cfg, sd, crops, ref, taps[f"block{cfg.layers - 1}"], vit
    vit.close()


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_vit_l14_336_matches_fp32_oracle(tower, gpu, precision):
    cfg, sd, crops, ref, x_last, vit = tower
    vit.set_precision(precision)
    try:
        emb = vit.encode(crops.to(gpu))
        again = vit.encode(crops.to(gpu))
        xl = vit.debug_run_layers(crops.to(gpu), cfg.layers).float().cpu()
    finally:
        vit.set_precision("bf16")
    assert emb.shape == (6, cfg.embed_dim) and torch.isfinite(emb).all()
    assert torch.equal(emb, again)                                           # bitwise repeatable
    omc = one_minus_cos(emb.cpu(), ref)
    print(f"{ARCH} {precision} 1-cos vs fp32 oracle:", omc)
    assert omc.max().item() < COS_TOL, omc
    rd = one_minus_cos(xl.flatten(1), x_last.flatten(1))                     # residual stream behind the LAST block, every token
    print(f"{ARCH} {precision} last-block residual 1-cos:", rd)
    # bf16: half the embedding budget.  e4m3 operands (no reference mode to match): the whole budget --
    # measured 5.6e-4 at 577 tokens (4.9-5.9e-4 at 257), i.e. the operand rounding of 96 e4m3 GEMMs, not a function of the token count
    assert rd.max().item() < (5e-4 if precision == "bf16" else COS_TOL), rd


def test_vit_l14_336_batch_properties(tower, gpu):
    """Size-independent properties on a batch the oracle cannot cover: 2 x 128 crops (147 712 token rows, ragged last M tile)."""
    cfg, _, _, _, _, vit = tower
    n = 256
    g = torch.Generator(device=gpu).manual_seed(9)
    crops = torch.randn(n, 3, cfg.image_size, cfg.image_size, device=gpu, generator=g)
    crops[200] = crops[3]
    e = vit.encode(crops)
    assert torch.isfinite(e).all() and torch.allclose(e.norm(dim=-1), torch.ones(n, device=gpu), atol=1e-5)
    assert torch.equal(e, vit.encode(crops)) and torch.equal(e[200], e[3])
    vit.set_chunk(100)                                                       # 100 + 100 + 56 crops per pass
    try:
        chunked = vit.encode(crops)
    finally:
        vit.set_chunk(2048)
    assert one_minus_cos(e.cpu(), chunked.cpu()).max().item() < 1e-6


def test_shipped_checkpoint_chain_on_its_own_tower(gpu, golden_dir, tmp_path):
    """What `predict_simple.py` runs with the artifact the reference ships: AestheticRegressor = ViT-L-14-336 tower ->
    'centre_crop' embedding -> the 768-264-128-64-1 SimpleFC of models/single_crop_regression_9.4k_imgs_80_epochs.pth
    (weights: tests/golden/regressor_shipped.npz, taken from that file by make_golden.py).  The tower's weights are seeded
    (no network: the openai checkpoint cannot be fetched); the regressor's are the real ones."""
    from clip_assisted_data_labeling_amd.embedder import AestheticRegressor
    from clip_assisted_data_labeling_amd.preprocess import ClipValTransform, extract_crops
    gs = np.load(os.path.join(golden_dir, "regressor_shipped.npz"))
    L = int(gs["n_layers"])
    Ws, bs = [gs[f"W{i}"] for i in range(L)], [gs[f"b{i}"] for i in range(L)]
    slope = float(gs["negative_slope"])
    sizes = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
    assert sizes == [768, 264, 128, 64, 1]
    model_name = f"{ARCH}/seed0"                                             # shipped: 'ViT-L-14-336/openai'
    m = SimpleFC(sizes[0], sizes[1:-1], 1, clip_models=[model_name], crop_names=["centre_crop"])
    with torch.no_grad():
        for layer, W, b in zip(m._linears(), Ws, bs):
            layer.weight.copy_(torch.from_numpy(W)); layer.bias.copy_(torch.from_numpy(b))
    torch.save(m, tmp_path / "shipped_like.pth")
    reg = AestheticRegressor(str(tmp_path / "shipped_like.pth"), device="cuda", verbose=0)
    assert reg.clip_models[0].img_resolution == 336
    rs = np.random.RandomState(5)
    imgs = [Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((400, 520), (336, 336), (700, 380))]
    scores, feats = reg.predict_scores(imgs)
    assert feats.shape == (3, 768) and scores.shape == (3,)
    # score = the C oracle's SimpleFC on the returned embedding (utils/nn_model.py:38-41)
    ref_score = fcreg_oracle.forward_c(Ws, bs, feats.cpu().numpy(), slope)[:, 0]
    assert np.abs(scores.cpu().numpy() - ref_score).max() < SCORE_TOL
    # and the embedding is the fp32 oracle's for the centre crop of the image
    cfg = vit_config.ARCHS[ARCH]
    sd = vit_config.seeded_state_dict(cfg, 0)
    crop0 = torch.stack([ClipValTransform(336)(extract_crops(im.convert("RGB"))[0][0]) for im in imgs[:2]])
    ref = vit_oracle.encode_image(sd, cfg, crop0)
    assert one_minus_cos(feats[:2].cpu(), ref).max().item() < COS_TOL
    one, f1 = reg.predict_score(imgs[0])
    assert abs(one - float(scores[0])) < 1e-5 and torch.equal(f1[0], feats[0])
