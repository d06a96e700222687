"""Similarity search: the oracle against the reference's own compute_distance / topN vectors, and
the driver's host logic with the device call replaced by the oracle."""
import os

import numpy as np
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import similar_driver
from oracle import simsearch_oracle


def test_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "simsearch_small.npz"))
    for measure in ("l2", "cosine"):
        d = simsearch_oracle.distances(g["emb"], g["query"], measure)
        assert np.abs(d - g[f"dist_{measure}"]).max() <= 2e-7
        idx, val = simsearch_oracle.top_n(g[f"dist_{measure}"], int(g["top_n"]))
        assert sorted(idx.tolist()) == g[f"kept_{measure}"].tolist()          # the set the reference's topN keeps
        assert np.all(np.diff(val) >= 0)
    assert g["dist_l2"][3] == g["dist_l2"][7]                                  # the planted tie: lower index first
    i, _ = simsearch_oracle.top_n(np.array([2.0, 1.0, np.nan, 1.0, 0.5], np.float32), 4)
    assert i.tolist() == [4, 1, 3, 0]
    i, v = simsearch_oracle.top_n(np.array([np.nan, 3.0], np.float32), 5)
    assert i.tolist() == [1, 0] and v[1] == np.inf


def _pt(path, vec, model="M/x"):
    torch.save({model: {"square_padded_crop": torch.tensor(vec, dtype=torch.float32).unsqueeze(0)}}, path)


def test_driver_host_logic_with_oracle_backend(tmp_path, monkeypatch):
    def fake_nearest(emb, query, measure, top_n, device="cuda", **kw):
        d = simsearch_oracle.distances(emb.numpy().reshape(emb.shape[0], -1), query, measure)
        return simsearch_oracle.top_n(d, top_n)
    monkeypatch.setattr(similar_driver, "nearest", fake_nearest)
    ctx, search = tmp_path / "ctx", tmp_path / "search"
    os.makedirs(ctx); os.makedirs(search / "deep")
    _pt(ctx / "c0.pt", [1.0, 0.0, 0.0]); _pt(ctx / "c1.pt", [0.8, 0.2, 0.0])
    torch.save("garbage", ctx / "broken.pt")                                   # skipped with a message (:52-55)
    vecs = {"s0": [0.9, 0.1, 0.0], "deep/s1": [0.0, 1.0, 0.0], "s2": [0.85, 0.15, 0.05], "s3": [-1.0, 0.0, 0.0], "nojpg": [0.9, 0.1, 0.0]}
    for k, v in vecs.items():
        _pt(search / (k + ".pt"), v)
        if k != "nojpg":
            Image.new("RGB", (8, 8)).save(search / (k + ".jpg"))
    _pt(search / "c0.pt", [0.9, 0.1, 0.0]); Image.new("RGB", (8, 8)).save(search / "c0.jpg")   # same name as a context image
    args = similar_driver.argparse.Namespace(context_dir=str(ctx), search_dir=str(search), output_dir=None,
                                             clip_models_to_use=["all"], crop_name_to_use="square_padded_crop",
                                             similarity_measure="cosine", top_n=3, packed_store=None)
    emb, names = similar_driver.create_context_embedding(args, str(ctx))
    assert args.clip_models_to_use == ["M/x"] and sorted(names) == ["c0.pt", "c1.pt"]
    assert np.allclose(emb, [0.9, 0.1, 0.0])
    got = similar_driver.find_similar_imgs(args, emb, names)
    assert [os.path.relpath(p, search) for p, _ in got] == ["s0.jpg", "s2.jpg", os.path.join("deep", "s1.jpg")]
    assert got[0][1] < 1e-6 and got[0][1] <= got[1][1] <= got[2][1]
    similar_driver.main(["--context_dir", str(ctx), "--search_dir", str(search), "--similarity_measure", "l2", "--top_n", "2"])
    out = sorted(os.listdir(ctx / "_similar"))
    assert len(out) == 2 and out[0].endswith("_s0.jpg") and out[0].startswith("0.000") and out[1].endswith("_s2.jpg")
