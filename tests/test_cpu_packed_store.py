"""Packed embedding store: lossless both ways against the reference's per-image `.pt` format,
resume and overwrite rules, and the feature assembly the regressor driver reads from it."""
import json
import os

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import embed_driver, predict_driver
from clip_assisted_data_labeling_amd.packed_store import (PackedStore, PackedStoreWriter, export_pt, image_key, import_pt)
from clip_assisted_data_labeling_amd.preprocess import CROP_NAMES
from tests.test_cpu_drivers import FakeEncoder, _make_images


def test_writer_reader_roundtrip_rotation_and_override(tmp_path):
    sd = str(tmp_path / "store")
    rs = np.random.RandomState(0)
    a = rs.randn(10, 4, 8).astype(np.float32)
    with PackedStoreWriter(sd, "M/x", CROP_NAMES, 8, rank=0, rotate_every=4) as w:
        w.append([f"k{i}" for i in range(7)], torch.from_numpy(a[:7]))
        w.append([f"sub/k{i}" for i in range(7, 10)], a[7:])
    assert len([f for f in os.listdir(sd) if f.endswith(".json")]) == 3         # 4 + 4 + 2 images
    st = PackedStore(sd)
    assert st.models() == ["M/x"]
    keys, data, crops = st.load("M/x")
    assert keys == [f"k{i}" for i in range(7)] + [f"sub/k{i}" for i in range(7, 10)] and crops == CROP_NAMES
    assert np.array_equal(np.asarray(data), a)                                   # bit-exact
    # a later writer (other rank, --force_reencode) overrides k3 and adds k10; key order stays first-seen
    b = rs.randn(2, 4, 8).astype(np.float32)
    with PackedStoreWriter(sd, "M/x", CROP_NAMES, 8, rank=1) as w:
        w.append(["k3", "k10"], b)
    keys2, data2, _ = PackedStore(sd).load("M/x")
    assert keys2 == keys + ["k10"]
    assert np.array_equal(data2[3], b[0]) and np.array_equal(data2[10], b[1]) and np.array_equal(data2[4], a[4])
    assert PackedStore(sd).keys("M/x") == set(keys2)
    with pytest.raises(ValueError):
        PackedStoreWriter(sd, "M/x", CROP_NAMES, 8).append(["z"], np.zeros((1, 3, 8), np.float32))
    with pytest.raises(KeyError):
        PackedStore(sd).load("other/model")


def test_unsealed_or_truncated_shards_are_not_trusted(tmp_path):
    sd = str(tmp_path / "store")
    w = PackedStoreWriter(sd, "M/x", ["a"], 4)
    w.append(["k0"], np.ones((1, 1, 4), np.float32))                             # never closed: no index yet
    assert PackedStore(sd).models() == []
    w.close()
    assert PackedStore(sd).keys("M/x") == {"k0"}
    data = [f for f in os.listdir(sd) if f.endswith(".f32")][0]
    open(os.path.join(sd, data), "wb").write(b"\0" * 8)                          # shorter than the index claims
    with pytest.raises(ValueError):
        PackedStore(sd)


def test_embed_driver_packed_store_equals_pt_files_and_resumes(tmp_path):
    root, root2 = str(tmp_path / "data"), str(tmp_path / "data2")
    os.makedirs(root)
    paths = _make_images(root, 7)
    sd = str(tmp_path / "store")
    enc = FakeEncoder()
    ds = embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=enc, device="cpu", packed_store=sd)
    assert ds.process() == (7, 0, 1)
    assert not any(f.endswith(".pt") for _, _, fs in os.walk(root) for f in fs)  # nothing written next to the images
    calls = enc.calls
    again = embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=enc, device="cpu", packed_store=sd)
    assert again.process()[:2] == (0, 7) and enc.calls == calls                 # resume from the store index
    # the classic per-image writer on a copy of the data set gives the same numbers
    import shutil
    shutil.copytree(root, root2)
    embed_driver.Feature_Dataset(root2, "Fake-A/test", 3, shuffle_filenames=False, encoder=FakeEncoder(), device="cpu").process()
    assert export_pt(sd, root) == 7
    for p in paths:
        a = torch.load(os.path.splitext(p)[0] + ".pt", weights_only=True)
        b = torch.load(os.path.splitext(p.replace(root, root2))[0] + ".pt", weights_only=True)
        assert list(a) == list(b) == ["Fake-A/test"] and list(a["Fake-A/test"]) == CROP_NAMES
        for c in CROP_NAMES:
            assert a["Fake-A/test"][c].shape == (1, 3) and a["Fake-A/test"][c].dtype == torch.float32
            assert torch.equal(a["Fake-A/test"][c], b["Fake-A/test"][c])
    # export merges with what is already in a file, import reads it all back
    embed_driver.Feature_Dataset(root, "Fake-B/test", 4, shuffle_filenames=False, encoder=enc, device="cpu").process()
    assert export_pt(sd, root) == 7
    assert set(torch.load(os.path.splitext(paths[0])[0] + ".pt", weights_only=True)) == {"Fake-A/test", "Fake-B/test"}
    sd2 = str(tmp_path / "store2")
    assert import_pt(root, sd2) == {"Fake-A/test": 7, "Fake-B/test": 7}
    k1, d1, _ = PackedStore(sd).load("Fake-A/test")
    k2, d2, _ = PackedStore(sd2).load("Fake-A/test")
    assert sorted(k1) == sorted(k2)
    assert np.array_equal(np.asarray(d1)[np.argsort(k1)], np.asarray(d2)[np.argsort(k2)])
    assert image_key(os.path.join(root, "sub", "img001.jpg"), root) == "sub/img001"


def test_store_features_match_pt_feature_assembly(tmp_path):
    sd = str(tmp_path / "store")
    E = 4
    crops = list(CROP_NAMES)
    with PackedStoreWriter(sd, "M1/x", crops, E) as w1, PackedStoreWriter(sd, "M2/y", crops, E) as w2:
        for i, k in enumerate(["u0", "u1", "u2"]):
            w1.append([k], np.arange(len(crops) * E, dtype=np.float32).reshape(1, len(crops), E) + 100 * i)
            if k != "u1":
                w2.append([k], -np.arange(len(crops) * E, dtype=np.float32).reshape(1, len(crops), E) - 100 * i)
    root = str(tmp_path / "data")
    os.makedirs(root)
    export_pt(sd, root)
    want_crops = ["subcrop2", "centre_crop"]
    store = PackedStore(sd)
    found, mat = store.features(["M2/y", "M1/x"], want_crops, ["u0", "u1", "u2", "nope"])
    assert found.tolist() == [True, False, True, False]                          # u1 lacks M2/y
    assert mat.shape == (2, 2 * 2 * E)
    for row, k in zip(mat, ["u0", "u2"]):
        ref = predict_driver.assemble_features(os.path.join(root, k + ".pt"), ["M2/y", "M1/x"], want_crops)
        assert torch.equal(torch.from_numpy(row.copy()), ref)
    # a crop the store does not hold is an error naming it, never a silently narrower feature row
    with pytest.raises(ValueError, match="missing_crop"):
        store.features(["M1/x"], ["centre_crop", "missing_crop"], ["u0"])
    # the key index is built once per store object and reused by every call (predict_driver calls features() per batch)
    idx = store._index("M1/x")
    store.features(["M1/x"], want_crops, ["u2"])
    assert store._index("M1/x") is idx


def test_train_set_from_store_equals_pt_path_and_rejects_unknown_crops(tmp_path):
    """train_driver --packed_store must assemble the rows the .pt path assembles, and a crop list the store cannot
    satisfy (the reference's stale default 'subcrop2_0.1', _4_train_model.py:266-267) must fail loudly instead of
    training on fewer crops than `model.crop_names` will claim."""
    import argparse
    import pandas as pd
    from clip_assisted_data_labeling_amd import train_driver
    E, crops = 4, list(CROP_NAMES)
    root = tmp_path / "data"
    (root / "setA").mkdir(parents=True)
    sd = str(tmp_path / "store")
    rs = np.random.RandomState(0)
    uuids = [f"img{i:02d}" for i in range(9)]
    with PackedStoreWriter(sd, "M1/x", crops, E, rank=0) as w0, PackedStoreWriter(sd, "M1/x", crops, E, rank=1) as w1:
        for i, u in enumerate(uuids):                              # two ranks -> two shards
            (w0 if i % 2 == 0 else w1).append([f"setA/{u}"], rs.randn(1, len(crops), E).astype(np.float32))
    export_pt(sd, str(root))
    pd.DataFrame({"uuid": uuids + ["ghost"], "label": list(np.linspace(0, 0.9, 9)) + [0.5],
                  "timestamp": 0}).to_csv(root / "setA.csv", index=False)
    base = dict(train_data_dir=str(root), train_data_names=["setA"], clip_models_to_use=["M1/x"], random_seed=3)
    want = ["centre_crop", "subcrop2"]
    f_pt, l_pt = train_driver.load_training_set(argparse.Namespace(packed_store=None, **base), want)
    f_st, l_st = train_driver.load_training_set(argparse.Namespace(packed_store=sd, **base), want)
    assert f_pt.shape == (9, 2 * E) and torch.equal(f_pt, f_st) and torch.equal(l_pt, l_st)
    with pytest.raises(ValueError, match="subcrop2_0.1"):
        train_driver.load_training_set(argparse.Namespace(packed_store=sd, **base), ["centre_crop", "subcrop2_0.1"])
    with pytest.raises(RuntimeError):                               # the .pt path skips every sample ('Missing crops')
        train_driver.load_training_set(argparse.Namespace(packed_store=None, **base), ["centre_crop", "subcrop2_0.1"])
