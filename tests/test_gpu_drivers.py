"""End-to-end drivers on the GPU: embed -> .pt store -> predict -> CSV/JSON/previews; dedup on the store."""
import json
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import dedup_driver, embed_driver, predict_driver, vit_config
from clip_assisted_data_labeling_amd.nn_model import SimpleFC
from clip_assisted_data_labeling_amd.preprocess import CROP_NAMES, ClipValTransform, extract_crops
from oracle import fcreg_oracle, vit_oracle
from tests.helpers import np_fc_weights, one_minus_cos

pytestmark = pytest.mark.gpu
MODEL = "ViT-small-test/seed4"


def _dataset(root, n):
    rs = np.random.RandomState(3)
    os.makedirs(root)
    for i in range(n):
        w, h = rs.randint(100, 200), rs.randint(100, 200)
        arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        if i == n - 1:                                  # exact duplicate of image 0 under another name
            arr = np.asarray(Image.open(os.path.join(root, "a000.jpg")))
        Image.fromarray(arr).save(os.path.join(root, f"a{i:03d}.jpg"), quality=95)
    with open(os.path.join(root, "a001.json"), "w") as f:
        json.dump({"prompt": "x"}, f)


def test_embed_predict_dedup_pipeline(gpu, tmp_path):
    root = str(tmp_path / "imgs")
    _dataset(root, 9)
    ds = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda")
    assert ds.process() == (9, 0, 0)
    cfg = vit_config.config_for(MODEL)
    sd = vit_config.seeded_state_dict(cfg, 4)
    # stored embeddings == oracle on the same crops
    img = Image.open(os.path.join(root, "a003.jpg")).convert("RGB")
    crops, names = extract_crops(img)
    ref = vit_oracle.encode_image(sd, cfg, torch.stack([ClipValTransform(cfg.image_size)(c) for c in crops]))
    d = torch.load(os.path.join(root, "a003.pt"), weights_only=True)[MODEL]
    got = torch.cat([d[c] for c in CROP_NAMES])
    assert got.shape == (4, cfg.embed_dim) and one_minus_cos(got, ref).max().item() < 1e-3

    # the same store written with the GPU front end (workers only decode) is identical, file by file
    ds2 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda",
                                       force_reencode=True, gpu_preprocess=True)
    before = {f: torch.load(os.path.join(root, f), weights_only=True) for f in sorted(os.listdir(root)) if f.endswith(".pt")}
    assert ds2.process() == (9, 0, 0)
    for f, old in before.items():
        new = torch.load(os.path.join(root, f), weights_only=True)
        assert all(torch.equal(new[MODEL][c], old[MODEL][c]) for c in CROP_NAMES), f

    # ... and so is the store written with the JPEG decode on the GPU as well (no decoding worker at all), including the files the
    # device decoder hands to Pillow: a progressive JPEG and a PNG are added for this run
    Image.open(os.path.join(root, "a002.jpg")).save(os.path.join(root, "b_prog.jpg"), quality=90, progressive=True)
    Image.open(os.path.join(root, "a004.jpg")).save(os.path.join(root, "b_png.png"))
    ref_run = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                           gpu_preprocess=True)
    assert ref_run.process() == (11, 0, 0)
    before = {f: torch.load(os.path.join(root, f), weights_only=True) for f in sorted(os.listdir(root)) if f.endswith(".pt")}
    assert len(before) == 11
    ds3 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                       gpu_decode=True, decode_chunk=5)
    ds3.gpu_decode_max_bytes = sorted(os.path.getsize(os.path.join(root, f)) for f in os.listdir(root) if f.endswith(".jpg"))[-3]
    assert ds3.process() == (11, 0, 0)                      # (the two largest files take the reader threads' Pillow route)
    for f, old in before.items():
        new = torch.load(os.path.join(root, f), weights_only=True)
        assert all(torch.equal(new[MODEL][c], old[MODEL][c]) for c in CROP_NAMES), f
    ds4 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                       gpu_decode=True)
    ds4.gpu_decode_progressive = True                       # ... and with the progressive file decoded on the device too
    assert ds4.process() == (11, 0, 0)
    # unreadable files are counted and skipped, like the reference's loader does (utils/embedder.py:176-181): an empty file, half a
    # file (the device flags its entropy data, Pillow then refuses it too), bytes that are no image at all
    good = open(os.path.join(root, "a005.jpg"), "rb").read()
    for name, blob in (("c_empty.jpg", b""), ("c_half.jpg", good[: len(good) // 2]), ("c_text.jpg", b"not an image" * 20)):
        with open(os.path.join(root, name), "wb") as f:
            f.write(blob)
    ds5 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                       gpu_decode=True)
    assert ds5.process() == (11, 0, 3)
    for name in ("c_empty", "c_half", "c_text"):
        assert not os.path.exists(os.path.join(root, name + ".pt"))
        os.remove(os.path.join(root, name + ".jpg"))
    for f, old in before.items():
        new = torch.load(os.path.join(root, f), weights_only=True)
        assert all(torch.equal(new[MODEL][c], old[MODEL][c]) for c in CROP_NAMES), f
    for f in ("b_prog.jpg", "b_prog.pt", "b_png.png", "b_png.pt"):
        os.remove(os.path.join(root, f))

    # a tiny pixel budget (every decode chunk is one or two images) and a device decoder that reports "no memory" for every other
    # group (those files then take the Pillow route): the same store
    ds6 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                       gpu_decode=True)
    ds6.gpu_decode_max_pixels = 30_000                      # images are 100..200 pixels a side
    calls, real_run = [], ds6.jpeg.lib.jpegdec_run

    class _Lib:                                             # the loaded library with one entry point wrapped
        def __getattr__(self, name):
            return getattr(ds6_lib, name)

        def jpegdec_run(self, *a):
            calls.append(1)
            return ds6.jpeg.NO_MEMORY if len(calls) % 2 == 0 else real_run(*a)
    ds6_lib, ds6.jpeg.lib = ds6.jpeg.lib, _Lib()
    assert ds6.process() == (9, 0, 0) and len(calls) >= 5
    ds6.jpeg.lib = ds6_lib
    for f, old in before.items():
        if f in ("b_prog.pt", "b_png.pt"):
            continue
        new = torch.load(os.path.join(root, f), weights_only=True)
        assert all(torch.equal(new[MODEL][c], old[MODEL][c]) for c in CROP_NAMES), f

    # --precision fp8: the e4m3 encoder behind the same loader (close to the bf16 store, not the same bits), then back to bf16
    ds7 = embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                       gpu_decode=True, precision="fp8")
    assert ds7.encoder.precision == "fp8" and ds7.process() == (9, 0, 0)
    same = 0
    for f, old in before.items():
        if f in ("b_prog.pt", "b_png.pt"):
            continue
        new = torch.load(os.path.join(root, f), weights_only=True)
        for c in CROP_NAMES:
            cos = torch.nn.functional.cosine_similarity(new[MODEL][c].double(), old[MODEL][c].double(), dim=-1).min().item()
            assert 1 - cos < 5e-3, (f, c, 1 - cos)
            same += int(torch.equal(new[MODEL][c], old[MODEL][c]))
    assert same == 0
    with pytest.raises(ValueError):
        embed_driver.Feature_Dataset(root, MODEL, 4, num_workers=0, device="cuda", precision="int4")
    assert embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda", force_reencode=True,
                                        gpu_decode=True).process() == (9, 0, 0)

    # regressor checkpoint in the reference's pickle format, then the predict driver
    sizes = [2 * cfg.embed_dim, 32, 16, 8, 1]
    Ws, bs = np_fc_weights(sizes, 5)
    m = SimpleFC(sizes[0], sizes[1:-1], 1, clip_models=[MODEL], crop_names=["centre_crop", "subcrop1"], dropout_prob=0.3)
    with torch.no_grad():
        for layer, W, b in zip(m._linears(), Ws, bs):
            layer.weight.copy_(torch.from_numpy(W)); layer.bias.copy_(torch.from_numpy(b))
    os.makedirs(tmp_path / "models")
    torch.save(m, tmp_path / "models" / "unit_test_regressor.pth")
    args = types.SimpleNamespace(root_dir=root, model_file=str(tmp_path / "models" / "unit_test_regressor.pth"),
                                 batch_size=4, copy_imgs_fraction=1.0, num_workers=0)
    db = predict_driver.predict_labels(args)
    csv = pd.read_csv(str(tmp_path / "imgs.csv"))
    assert list(csv.columns) == ["uuid", "label", "timestamp", "predicted_label"] and len(csv) == 9
    for uuid in ("a000", "a005"):
        dd = torch.load(os.path.join(root, uuid + ".pt"), weights_only=True)[MODEL]
        feat = torch.cat([dd["centre_crop"], dd["subcrop1"]], 0).flatten().numpy()[None]
        want = fcreg_oracle.forward_c(Ws, bs, feat)[0, 0]
        have = float(csv.loc[csv.uuid == uuid, "predicted_label"].iloc[0])
        assert abs(have - want) < 1e-4
    assert json.load(open(os.path.join(root, "a001.json")))["predicted_label"] == pytest.approx(
        float(csv.loc[csv.uuid == "a001", "predicted_label"].iloc[0]), abs=1e-6)
    previews = os.listdir(root + "_predicted_scores")
    assert len(previews) == 9 and all(p[5] == "_" and p.endswith(".jpg") for p in previews)
    # second run updates in place (new value overrides), no duplicate rows
    predict_driver.predict_labels(args)
    assert len(pd.read_csv(str(tmp_path / "imgs.csv"))) == 9

    # dedup on the store: (a seeded-random tiny tower maps noise images close together, so the threshold is
    # set from the data) the planted duplicate (a000, a008) must be reported, in row-major (i < j) order
    embs = torch.stack([torch.load(os.path.join(root, f"a{i:03d}.pt"), weights_only=True)[MODEL]["square_padded_crop"][0]
                        for i in range(9)]).half().float()
    embs = embs / embs.norm(dim=1, keepdim=True)
    sim = (embs @ embs.T).fill_diagonal_(0)
    assert sim[0, 8] == torch.triu(sim, 1).max()              # (upper triangle: a CPU matmul's [0, 8] and [8, 0] may differ in the last bit)
    others = sim.clone(); others[0, 8] = others[8, 0] = 0
    thr = float((others.max() + sim[0, 8]) / 2)
    dargs = types.SimpleNamespace(root_dir=root, threshold=thr, mode="copy", clip_model_to_use=None, chunk_size=10000, test=True)
    found = dedup_driver.find_near_duplicates(dargs)
    assert [(os.path.basename(a), os.path.basename(b)) for a, b, _ in found] == [("a000.jpg", "a008.jpg")]
    assert abs(found[0][2] - float(sim[0, 8])) < 1e-3
    dargs.test = False
    dedup_driver.find_near_duplicates(dargs)
    out = os.listdir(str(tmp_path / f"near_duplicates_cosine_{thr}"))
    assert any("_source_a000.jpg" in f for f in out) and any("_target_a008.pt" in f for f in out)


def test_pipeline_through_packed_store_equals_pt_pipeline(gpu, tmp_path):
    """embed --packed_store -> predict / dedup reading the shards == the per-image .pt pipeline, value for value."""
    from clip_assisted_data_labeling_amd.packed_store import PackedStore, export_pt
    root, root_pt = str(tmp_path / "imgs"), str(tmp_path / "ref" / "imgs")
    _dataset(root, 9)
    import shutil
    os.makedirs(tmp_path / "ref")
    shutil.copytree(root, root_pt)
    sd = str(tmp_path / "store")
    assert embed_driver.Feature_Dataset(root, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda",
                                        packed_store=sd).process() == (9, 0, 0)
    assert embed_driver.Feature_Dataset(root_pt, MODEL, 4, shuffle_filenames=False, num_workers=0, device="cuda").process() == (9, 0, 0)
    keys, data, crops = PackedStore(sd).load(MODEL)
    assert sorted(keys) == [f"a{i:03d}" for i in range(9)] and crops == CROP_NAMES
    for k, row in zip(keys, np.asarray(data)):
        d = torch.load(os.path.join(root_pt, k + ".pt"), weights_only=True)[MODEL]
        assert torch.equal(torch.from_numpy(row.copy()), torch.cat([d[c] for c in CROP_NAMES]))     # bitwise

    cfg = vit_config.config_for(MODEL)
    sizes = [2 * cfg.embed_dim, 32, 16, 8, 1]
    Ws, bs = np_fc_weights(sizes, 5)
    m = SimpleFC(sizes[0], sizes[1:-1], 1, clip_models=[MODEL], crop_names=["centre_crop", "subcrop1"], dropout_prob=0.3)
    with torch.no_grad():
        for layer, W, b in zip(m._linears(), Ws, bs):
            layer.weight.copy_(torch.from_numpy(W)); layer.bias.copy_(torch.from_numpy(b))
    torch.save(m, tmp_path / "reg.pth")
    common = dict(model_file=str(tmp_path / "reg.pth"), batch_size=4, copy_imgs_fraction=0.0, num_workers=0)
    predict_driver.predict_labels(types.SimpleNamespace(root_dir=root, packed_store=sd, store_root=root, **common))
    predict_driver.predict_labels(types.SimpleNamespace(root_dir=root_pt, **common))
    a = pd.read_csv(str(tmp_path / "imgs.csv")).sort_values("uuid")
    b = pd.read_csv(str(tmp_path / "ref" / "imgs.csv")).sort_values("uuid")
    assert list(a.uuid) == list(b.uuid) and np.array_equal(a.predicted_label.values, b.predicted_label.values)

    common = dict(threshold=0.999, mode="copy", clip_model_to_use=None, chunk_size=10000, test=True)
    fa = dedup_driver.find_near_duplicates(types.SimpleNamespace(root_dir=root, packed_store=sd, **common))
    fb = dedup_driver.find_near_duplicates(types.SimpleNamespace(root_dir=root_pt, **common))
    assert [(os.path.basename(x), os.path.basename(y), v) for x, y, v in fa] == \
           [(os.path.basename(x), os.path.basename(y), v) for x, y, v in fb]
    assert ("a000.jpg", "a008.jpg") in [(os.path.basename(x), os.path.basename(y)) for x, y, _ in fa]
    assert export_pt(sd, root) == 9                         # and the reference's own scripts can still read the result
    assert torch.equal(torch.load(os.path.join(root, "a004.pt"), weights_only=True)[MODEL]["subcrop2"],
                       torch.load(os.path.join(root_pt, "a004.pt"), weights_only=True)[MODEL]["subcrop2"])


def test_aesthetic_regressor_fused_single_image(gpu, tmp_path):
    """The composition utils/embedder.py:298-311 intends: image -> crops -> encode -> [crop][E] -> score."""
    from clip_assisted_data_labeling_amd.embedder import AestheticRegressor
    from clip_assisted_data_labeling_amd.predict_simple import predict_images
    cfg = vit_config.config_for(MODEL)
    sd = vit_config.seeded_state_dict(cfg, 4)
    sizes = [3 * cfg.embed_dim, 40, 12, 1]
    Ws, bs = np_fc_weights(sizes, 9)
    m = SimpleFC(sizes[0], sizes[1:-1], 1, clip_models=[MODEL], crop_names=["subcrop2", "centre_crop", "square_padded_crop"])
    with torch.no_grad():
        for layer, W, b in zip(m._linears(), Ws, bs):
            layer.weight.copy_(torch.from_numpy(W)); layer.bias.copy_(torch.from_numpy(b))
    torch.save(m, tmp_path / "reg.pth")
    rs = np.random.RandomState(11)
    imgs = [Image.fromarray(rs.randint(0, 256, (rs.randint(120, 260), rs.randint(120, 260), 3), dtype=np.uint8)) for _ in range(3)]
    reg = AestheticRegressor(str(tmp_path / "reg.pth"), device="cuda", verbose=0)
    score, feats = reg.predict_score(imgs[1])
    assert isinstance(score, float) and feats.shape == (1, 3 * cfg.embed_dim)
    crops, names = extract_crops(imgs[1].convert("RGB"))
    ref = vit_oracle.encode_image(sd, cfg, torch.stack([ClipValTransform(cfg.image_size)(c) for c in crops]))
    ref_feat = torch.cat([ref[names.index(c)] for c in m.crop_names]).numpy()[None]      # model.crop_names order
    assert one_minus_cos(feats.cpu().view(3, -1), torch.from_numpy(ref_feat).view(3, -1)).max().item() < 1e-3
    assert abs(score - fcreg_oracle.forward_c(Ws, bs, feats.cpu().numpy())[0, 0]) < 1e-4
    batch_scores, _ = reg.predict_scores(imgs)
    assert abs(float(batch_scores[1]) - score) < 1e-5
    os.makedirs(tmp_path / "in")
    for i, im in enumerate(imgs):
        im.save(tmp_path / "in" / f"p{i}.png")
    res = predict_images([str(tmp_path / "in" / f"p{i}.png") for i in range(3)], str(tmp_path / "reg.pth"), "cuda",
                         str(tmp_path / "out"))
    assert len(res) == 3 and len(os.listdir(tmp_path / "out")) == 3
