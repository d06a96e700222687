"""BASELINE.json configs at their stated sizes (-m gpu), each through the C ABI and checked against the oracle:

* configs[2]  ViT-L/14 encode + fused regressor, 1024 images x 4 crops through `clipenc_encode_score`, with the
  3072-264-128-64-1 regressor of `regressor_4crop.npz` and with the reference's shipped single-crop checkpoint
  (`regressor_shipped.npz`, /root/reference/models/single_crop_regression_9.4k_imgs_80_epochs.pth);
* full width x full depth planted-outlier towers (what real CLIP statistics look like and seeded weights do not),
  bf16 and fp8 (configs[1], configs[3] arithmetic);
* configs[4]  cosine all-pairs dedup on 100 000 x 768 embeddings with 1 000 planted pairs
  (/root/reference/_2_remove_duplicates.py:63-80).

Full-size oracle runs are restricted to a handful of crops (the fp32 CPU tower takes ~0.5 s per crop); everything
else is checked through size-independent properties: scores are a pure function of the returned embeddings (checked
for ALL rows against the C oracle), bitwise repeatability, and row independence inside the 2048-crop chunk whose
last M-tile is ragged (2048 x 257 = 526 336 token rows = 2 056 tiles of 256).
"""
import os

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib, vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from clip_assisted_data_labeling_amd.nn_model import HipRegressor
from oracle import dedup_oracle, fcreg_oracle, vit_oracle
from tests.helpers import CLIP_MEAN, CLIP_STD, np_fc_weights, one_minus_cos, synthetic_crops

pytestmark = pytest.mark.gpu

COS_TOL = 1e-3      # north_star: embeddings within 1e-3 cosine of the fp32 CPU path
SCORE_TOL = 1e-4    # north_star: scores within 1e-4 abs


def _device_crops(n, size, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    u = torch.randint(0, 256, (n, 3, size, size), generator=g, device=dev, dtype=torch.int32).float()
    return ((u / 255.0 - CLIP_MEAN.to(dev)) / CLIP_STD.to(dev)).contiguous()


@pytest.fixture(scope="module")
def vit_l14(gpu):
    cfg = vit_config.ARCHS["ViT-L-14"]
    sd = vit_config.seeded_state_dict(cfg, 0)
    vit = HipViT(cfg, sd, gpu)
    yield cfg, sd, vit
    vit.close()


# ------------------------------------------------------------------------------------- configs[2]
def test_config2_vit_l14_encode_score_1024_images(vit_l14, gpu, golden_dir):
    cfg, sd, vit = vit_l14
    torch.set_num_threads(min(32, torch.get_num_threads()))
    n_img, E = 1024, cfg.embed_dim
    crops = _device_crops(n_img * 4, 224, 4242, gpu)                      # 4096 crops = two 2048-crop chunks
    # row independence: the same crop at the first row, at the LAST row of a chunk (inside the ragged 2 056th tile),
    # and in the second chunk must give the same bits
    crops[2047] = crops[5]
    crops[2048 + 1234] = crops[5]
    g4 = np.load(os.path.join(golden_dir, "regressor_4crop.npz"))
    Ws4, bs4 = np_fc_weights([int(s) for s in g4["sizes"]], int(g4["weight_seed"]))
    reg4 = HipRegressor([torch.from_numpy(w) for w in Ws4], [torch.from_numpy(b) for b in bs4], 0.01, gpu)
    assert np.abs(reg4(torch.from_numpy(g4["x"]).to(gpu)).cpu().numpy() - g4["y"]).max() < SCORE_TOL   # same weights as the fixture
    gs = np.load(os.path.join(golden_dir, "regressor_shipped.npz"))
    Ws1 = [gs[f"W{i}"] for i in range(int(gs["n_layers"]))]
    bs1 = [gs[f"b{i}"] for i in range(int(gs["n_layers"]))]
    reg1 = HipRegressor([torch.from_numpy(w) for w in Ws1], [torch.from_numpy(b) for b in bs1], float(gs["negative_slope"]), gpu)

    emb, score4 = vit.encode_score(crops, reg4, 4, [0, 1, 2, 3])
    torch.cuda.synchronize()
    assert emb.shape == (n_img, 4, E) and score4.shape == (n_img, 1)
    assert torch.isfinite(emb).all() and torch.isfinite(score4).all()
    assert torch.allclose(emb.norm(dim=-1), torch.ones(n_img, 4, device=gpu), atol=1e-5)
    flat = emb.view(-1, E)
    assert torch.equal(flat[5], flat[2047]) and torch.equal(flat[5], flat[2048 + 1234])
    # bitwise repeatability of the whole fused call at the benchmarked shape
    emb_b, score_b = vit.encode_score(crops, reg4, 4, [0, 1, 2, 3])
    assert torch.equal(emb, emb_b) and torch.equal(score4, score_b)
    # scores of ALL rows against the C oracle on the returned embeddings (utils/nn_model.py:38-41, [model][crop][E] order)
    emb_h = emb.cpu().numpy()
    ref4 = fcreg_oracle.forward_c(Ws4, bs4, emb_h.reshape(n_img, 4 * E))
    assert np.abs(score4.cpu().numpy() - ref4).max() < SCORE_TOL
    # Seeded towers map noise crops to nearly parallel embeddings, so the fixture regressor's scores barely move
    # (std ~2e-5).  A discriminating check: the same network with its first layer centred on the batch mean and its gain
    # raised until the scores spread over (0, 1) -- still `SimpleFC` arithmetic, oracle and kernel see the same weights.
    mean_feat = emb_h.reshape(n_img, 4 * E).mean(0)
    for gain in (30.0, 1000.0, 30000.0, 1e6):
        Wg = [Ws4[0] * gain] + Ws4[1:]
        bg = [(bs4[0] - Wg[0] @ mean_feat).astype(np.float32)] + bs4[1:]
        refg = fcreg_oracle.forward_c(Wg, bg, emb_h.reshape(n_img, 4 * E))
        if refg.std() > 0.02:
            break
    assert refg.std() > 0.02, refg.std()
    regg = HipRegressor([torch.from_numpy(w) for w in Wg], [torch.from_numpy(b) for b in bg], 0.01, gpu)
    _, scoreg = vit.encode_score(crops, regg, 4, [0, 1, 2, 3])
    assert np.abs(scoreg.cpu().numpy() - refg).max() < SCORE_TOL
    regg.close()
    # the reference's shipped single-crop checkpoint on the same batch: crop_names = ['centre_crop'] -> crop 0
    emb1, score1 = vit.encode_score(crops, reg1, 4, [0])
    assert torch.equal(emb1, emb)
    ref1 = fcreg_oracle.forward_c(Ws1, bs1, emb_h[:, 0, :], float(gs["negative_slope"]))
    assert np.abs(score1.cpu().numpy() - ref1).max() < SCORE_TOL
    # embeddings of 8 sampled crops (both chunks, first / last rows, the ragged tile) against the fp32 oracle
    idx = [0, 5, 777, 2046, 2047, 2048, 3001, 4095]
    ref = vit_oracle.encode_image(sd, cfg, crops[idx].cpu())
    omc = one_minus_cos(flat[idx].cpu(), ref)
    assert omc.max().item() < COS_TOL, omc
    # and the scores computed from the ORACLE's embeddings stay within the score tolerance scaled by the embedding
    # distance (the regressor is 1-Lipschitz-ish on unit vectors): end-to-end sanity of configs[2]
    img = [i // 4 for i in idx if i % 4 == 0]
    feats_ref = emb_h[img].copy()
    for i, r in zip(idx, ref.numpy()):
        if i // 4 in img:
            feats_ref[img.index(i // 4), i % 4] = r
    e2e = fcreg_oracle.forward_c(Ws4, bs4, feats_ref.reshape(len(img), 4 * E))
    assert np.abs(e2e - score4.cpu().numpy()[img]).max() < 5e-3
    reg4.close(); reg1.close()


# ----------------------------------------------------------------- real-statistics stand-in at full size
def _plant_outliers(sd, cfg, seed=0):
    """Residual channels 100-300x the median and rows with non-zero mean, at every depth (SURVEY.md §7 'precision
    budget': real CLIP ViT-L has them, seeded weights do not)."""
    g = torch.Generator().manual_seed(seed)
    hot = torch.randperm(cfg.width, generator=g)[:5]
    sd["ln_pre.weight"][hot] *= torch.tensor([50.0, 70.0, 90.0, 110.0, 40.0])
    sd["ln_pre.bias"] += 1.5                              # every row gets a mean of ~1.5 sigma
    sd["ln_pre.bias"][hot] += torch.tensor([60.0, -80.0, 100.0, -40.0, 50.0])
    for l in range(cfg.layers):
        sd[f"transformer.resblocks.{l}.attn.out_proj.bias"][hot[l % 5]] += 4.0
        sd[f"transformer.resblocks.{l}.mlp.c_proj.bias"][hot[(l + 2) % 5]] -= 4.0
    return hot


@pytest.fixture(scope="module")
def vit_l14_outliers(gpu):
    cfg = vit_config.ARCHS["ViT-L-14"]
    sd = vit_config.seeded_state_dict(cfg, 8)
    hot = _plant_outliers(sd, cfg)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    crops = synthetic_crops(6, 224, 31)
    taps = {}
    ref = vit_oracle.encode_image(sd, cfg, crops, taps)
    x_last = taps[f"block{cfg.layers - 1}"]
    med = x_last.abs().median()
    big = (x_last.abs().amax(dim=(0, 1)) > 100 * med).sum().item()
    assert big >= 4, (big, med)                                            # >= 4 channels at >= 100x the median, at the LAST block
    assert x_last.abs().max() < 400 * med
    assert (x_last.mean(-1).abs() / x_last.std(-1)).mean() > 0.05          # rows are not zero-mean
    vit = HipViT(cfg, sd, gpu)
    yield cfg, crops, ref, x_last, vit
    vit.close()


def test_vit_l14_full_depth_planted_outliers_bf16(vit_l14_outliers, gpu):
    """The LayerNorm fold subtracts mean x colsum AFTER a bf16 GEMM (gemm_persist.hip EPI_LNFOLD): cancellation grows
    with width and depth, so it is exercised here at 1024 x 24, not only on the 3-layer toy."""
    cfg, crops, ref, x_last, vit = vit_l14_outliers
    emb = vit.encode(crops.to(gpu)).cpu()
    omc = one_minus_cos(emb, ref)
    print("ViT-L/14 planted outliers bf16 1-cos:", omc)
    assert omc.max().item() < COS_TOL, omc
    xl = vit.debug_run_layers(crops.to(gpu), cfg.layers).float().cpu()
    rd = one_minus_cos(xl.flatten(1), x_last.flatten(1))
    print("last-block residual 1-cos:", rd)
    assert rd.max().item() < 5e-4, rd


def test_vit_l14_full_depth_planted_outliers_fp8(vit_l14_outliers, gpu):
    cfg, crops, ref, x_last, vit = vit_l14_outliers
    vit.set_precision("fp8")
    try:
        emb = vit.encode(crops.to(gpu)).cpu()
        xl = vit.debug_run_layers(crops.to(gpu), cfg.layers).float().cpu()
    finally:
        vit.set_precision("bf16")
    omc = one_minus_cos(emb, ref)
    print("ViT-L/14 planted outliers fp8 1-cos:", omc)
    assert omc.max().item() < COS_TOL, omc
    rd = one_minus_cos(xl.flatten(1), x_last.flatten(1))
    print("last-block residual 1-cos (fp8):", rd)
    assert rd.max().item() < 5e-4, rd


# ------------------------------------------------------------------------------------- configs[4]
def test_config4_dedup_100k_planted_pairs(gpu):
    """E fp16 [100 000, 768], rows ~ N(0,1), + 1 000 planted near-duplicates e_j = e_i + 0.1 noise (cos ~ 0.995; random
    pairs never exceed ~0.2), threshold 0.96: the pair set must be EXACTLY the planted set, the count exact
    (SURVEY.md §8d cfg5).  The reference caps a chunk at 10 000 rows (three N x N temporaries)."""
    lib = _lib.load()
    n, d, planted, thr = 100_000, 768, 1000, 0.96
    g = torch.Generator(device=gpu).manual_seed(7)
    e = torch.randn(n, d, device=gpu, generator=g)
    src = torch.randperm(n - planted, device=gpu, generator=g)[:planted]
    e[n - planted:] = e[src] + 0.1 * torch.randn(planted, d, device=gpu, generator=g)
    e16 = e.half().contiguous()
    n_pad, d_pad = (n + 255) // 256 * 256, (d + 127) // 128 * 128
    ws = torch.empty(n_pad * d_pad, dtype=torch.float16, device=gpu)
    cap = 1 << 16
    pairs = torch.full((cap, 2), -1, dtype=torch.int64, device=gpu)
    vals = torch.zeros(cap, dtype=torch.float32, device=gpu)
    count = torch.full((1,), 123, dtype=torch.int64, device=gpu)
    st = _lib.current_stream_ptr(gpu)
    for _ in range(2):                                                    # the second run must reproduce the first
        _lib.check(lib.dedup_find_pairs(e16.data_ptr(), n, d, thr, 1, ws.data_ptr(), pairs.data_ptr(), vals.data_ptr(), cap,
                                        count.data_ptr(), st), "dedup")
        torch.cuda.synchronize()
        c = int(count.item())
        assert c == planted, c
        p = pairs[:c].cpu().numpy()
        got = {tuple(r) for r in p.tolist()}
        want = {(int(s), n - planted + t) for t, s in enumerate(src.cpu().tolist())}
        assert got == want
    # values: the oracle's fp16 arithmetic (normalise in half, fp16 dot output) on the planted rows only
    v = vals[:c].cpu().numpy()
    rows = torch.cat([e16[p[:, 0]], e16[p[:, 1]]]).cpu()
    s32 = dedup_oracle.similarity_fp32(rows).numpy()
    ref = np.array([s32[k, c + k] for k in range(c)])
    assert np.abs(v - ref).max() <= 1.0e-3                                 # fp16 rounding of the normalised rows and of the output
    assert v.min() > thr and v.max() <= 1.0005


# ------------------------------------------------------------------------------------- configs[3]
@pytest.mark.parametrize("precision", ["fp8", "bf16"])
def test_config3_job_runner_small_job_on_the_gpu(vit_l14, gpu, precision):
    """BASELINE.json configs[3] control flow with the real encoder: a 600-image job in batches of 256 (ragged last batch of
    88), uint8 crops generated on the device per batch from the counter-based source, results kept in HBM.  Every stored row
    must equal what a stand-alone call returns for the same generated crops, and a second run of the job must reproduce the
    first bit for bit (deterministic encoder + reproducible source)."""
    from clip_assisted_data_labeling_amd.job import run_embed_job, synthetic_u8_source
    cfg, sd, vit = vit_l14
    Ws, bs = np_fc_weights([4 * cfg.embed_dim, 264, 128, 64, 1], 21)
    reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], 0.01, gpu)
    vit.set_precision(precision)
    try:
        def run():
            src = synthetic_u8_source(cfg.image_size, 4, 777, 0, gpu)
            return run_embed_job(600, 256, 4, cfg.embed_dim, 1, src, lambda c: vit.encode_score(c, reg, 4, [0, 1, 2, 3]), gpu,
                                 sync=torch.cuda.synchronize)
        a, b = run(), run()
        assert a["batches"] == 3 and a["emb"].shape == (600, 4, cfg.embed_dim) and a["score"].shape == (600, 1)
        assert torch.equal(a["emb"], b["emb"]) and torch.equal(a["score"], b["score"])
        assert torch.isfinite(a["emb"]).all() and torch.allclose(a["emb"].norm(dim=-1), torch.ones(600, 4, device=gpu), atol=1e-5)
        # the same crops outside the job: regenerate the stream and encode the ragged LAST batch on its own
        src = synthetic_u8_source(cfg.image_size, 4, 777, 0, gpu)
        src(0, 256); src(256, 256)
        last = src(512, 88)
        e, s = vit.encode_score(last, reg, 4, [0, 1, 2, 3])
        assert torch.equal(e, a["emb"][512:]) and torch.equal(s, a["score"][512:])
        # scores are the C oracle's on the stored embeddings (utils/nn_model.py:38-41)
        ref = fcreg_oracle.forward_c(Ws, bs, a["emb"].cpu().numpy().reshape(600, -1))
        assert np.abs(a["score"].cpu().numpy() - ref).max() < SCORE_TOL
    finally:
        vit.set_precision("bf16")
        reg.close()
