"""Regressor training on the GPU (fctrain_*) against the reference's own SimpleFC + torch Adam + scheduler run
(tests/golden/train_small.npz, dropout 0) and against the oracle with dropout."""
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch

from clip_assisted_data_labeling_amd import train_driver
from clip_assisted_data_labeling_amd.nn_model import load_regressor
from clip_assisted_data_labeling_amd.train_driver import FcTrainer, cosine_warm_restarts_lr
from oracle import fcreg_oracle, train_oracle

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_small.npz"))
    L = len(g["hidden"]) + 1
    return g, [g[f"W0_{i}"] for i in range(L)], [g[f"b0_{i}"] for i in range(L)], [g[f"Wf_{i}"] for i in range(L)], [g[f"bf_{i}"] for i in range(L)]


def test_training_matches_reference_run(gpu, golden_dir):
    g, W0, b0, Wf, bf = _golden(golden_dir)
    tr = FcTrainer([torch.from_numpy(w) for w in W0], [torch.from_numpy(b) for b in b0], 0.01, gpu)
    X, T = torch.from_numpy(g["X"]).to(gpu), torch.from_numpy(g["T"]).to(gpu)
    for ep in range(len(g["orders"])):
        lr = cosine_warm_restarts_lr(float(g["lr"]), float(g["min_lr"]), int(g["T_0"]), ep)
        assert abs(lr - g["lrs"][ep]) < 1e-12                                   # the scheduler of the reference run
        losses = tr.epoch(X, T, g["orders"][ep], int(g["batch_size"]), lr, float(g["weight_decay"]))
        assert abs(float(losses.mean()) - g["losses"][ep]) < 2e-5, (ep, float(losses.mean()), g["losses"][ep])
    Ws, bs = tr.parameters()
    for a, b in zip(Ws + bs, Wf + bf):
        assert np.abs(a.numpy() - b).max() < 1e-3                                # 91 Adam steps in fp32 on both sides
    # eval-mode forward of the trained parameters == the oracle forward
    y = tr.predict(X).cpu().numpy()
    ref = fcreg_oracle.forward_np([w.numpy() for w in Ws], [b.numpy() for b in bs], g["X"])[:, 0]
    assert np.abs(ref - y).max() < 1e-5
    tr.close()


def test_training_with_dropout_matches_oracle(gpu, golden_dir):
    g, W0, b0, _, _ = _golden(golden_dir)
    p, seed, bs, wd = 0.4, 1234, 24, 1e-3
    X, T = g["X"], g["T"]
    orc = train_oracle.Trainer(W0, b0, wd, p, seed)
    tr = FcTrainer([torch.from_numpy(w) for w in W0], [torch.from_numpy(b) for b in b0], 0.01, gpu)
    Xd, Td = torch.from_numpy(X).to(gpu), torch.from_numpy(T).to(gpu)
    rs = np.random.RandomState(0)
    for ep in range(3):
        order = rs.permutation(len(X))[:190]                                     # a ragged last batch (190 = 7 x 24 + 22)
        ref = [orc.step(X[order[b0_:b0_ + bs]], T[order[b0_:b0_ + bs]], 1e-3) for b0_ in range(0, len(order), bs)]
        got = tr.epoch(Xd, Td, order, bs, 1e-3, wd, p, seed).cpu().numpy()
        assert np.abs(got - np.array(ref)).max() < 2e-5, (ep, np.abs(got - np.array(ref)).max())
    Ws, bs_ = tr.parameters()
    for a, b in zip(Ws + bs_, orc.W + orc.b):
        assert np.abs(a.numpy() - b).max() < 1e-3
    # dropout really was active: the same schedule without it gives different losses
    tr2 = FcTrainer([torch.from_numpy(w) for w in W0], [torch.from_numpy(b) for b in b0], 0.01, gpu)
    l0 = tr2.epoch(Xd, Td, rs.permutation(len(X)), bs, 1e-3, wd, 0.0, seed)
    l1 = FcTrainer([torch.from_numpy(w) for w in W0], [torch.from_numpy(b) for b in b0], 0.01, gpu).epoch(Xd, Td, None, bs, 1e-3, wd, p, seed)
    assert torch.isfinite(l0).all() and torch.isfinite(l1).all()
    tr.close(); tr2.close()


def test_train_driver_end_to_end_learns_and_saves_a_loadable_model(gpu, tmp_path, monkeypatch):
    rs = np.random.RandomState(1)
    n, E = 600, 32
    root = tmp_path / "data"
    os.makedirs(root / "setA")
    w = rs.randn(2 * E) / np.sqrt(2 * E)
    rows = []
    for i in range(n):
        e = rs.randn(2, E).astype(np.float32)
        label = float(e.reshape(-1) @ w) * 3 + 5 + 0.05 * rs.randn()
        torch.save({"M/x": {"centre_crop": torch.from_numpy(e[0:1].copy()), "subcrop2_0.1": torch.from_numpy(e[1:2].copy()),
                            "square_padded_crop": torch.zeros(1, E)}}, root / "setA" / f"u{i:04d}.pt")
        rows.append({"uuid": f"u{i:04d}", "label": label if i % 50 else np.nan, "timestamp": 0})
    pd.DataFrame(rows).to_csv(root / "setA.csv", index=False)
    monkeypatch.chdir(tmp_path)
    args = types.SimpleNamespace(train_data_dir=str(root), train_data_names=["setA"], model_name="unit", dont_save=False,
                                 clip_models_to_use=["all"], test_fraction=0.25, n_epochs=30, batch_size=16, lr=2e-3, min_lr=1e-6,
                                 restart_epochs=10, weight_decay=1e-4, dropout_prob=0.1, hidden_sizes=[32, 16], print_network_layout=False,
                                 random_seed=42, packed_store=None)
    model, losses, lrs, path = train_driver.train(args, ["centre_crop", "subcrop2_0.1"])
    assert len(losses[0]) == 30 and losses[1][-1] < 0.5 * losses[1][0] and losses[1][-1] < 0.01      # it learns the planted relation
    assert abs(lrs[0] - cosine_warm_restarts_lr(2e-3, 1e-6, 10, 1)) < 1e-12 and abs(lrs[9] - 2e-3) < 1e-12   # restart after 10 epochs
    assert path and os.path.exists(path) and "0.4k_imgs_30_epochs" in path               # 588 labelled -> 441 train
    # the artifact names the reference's class and loads back into the HIP-backed SimpleFC
    assert b"utils.nn_model" in open(path, "rb").read()
    m2 = load_regressor(path)
    assert m2.clip_models == ["M/x"] and m2.crop_names == ["centre_crop", "subcrop2_0.1"] and not m2.training
    x = torch.randn(5, 2 * E, device=gpu)
    assert torch.allclose(m2(x), model(x))
