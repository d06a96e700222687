"""Diversity ordering (reference _3_label_images.py:128-177): the oracle's semantics and the driver's host logic."""
import os
import random

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import diversity_driver
from oracle import diversity_oracle


def _unit(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, d, generator=g)
    return (e / e.norm(dim=1, keepdim=True)).numpy()


def test_oracle_is_the_sampled_farthest_point_walk():
    emb = _unit(60, 16, 0)
    samples = diversity_oracle.draw_samples(60, 12, 10, seed=3)
    order = diversity_oracle.diversity_order(emb, samples, first=0)
    # brute force in float64: candidate with the smallest largest cosine to the chosen set
    e = emb.astype(np.float64)
    chosen = [0]
    for t, s in enumerate(samples):
        mx = np.max(e[chosen] @ e[s].T, axis=0)
        assert int(s[int(np.argmin(mx))]) == order[t]
        chosen.append(order[t])
    # an already chosen image has cosine 1 with itself and is never picked again while anything else is on offer
    assert len(set(order)) == len(order) and 0 not in order


def test_oracle_ties_go_to_the_first_candidate_like_torch_argmin():
    emb = _unit(8, 4, 1)
    emb[5] = emb[2]                                                   # two identical candidates
    order = diversity_oracle.diversity_order(emb, [[5, 2, 0]], first=0)
    assert order == [5]
    order = diversity_oracle.diversity_order(emb, [[2, 5, 0]], first=0)
    assert order == [2]


def test_samples_are_the_positions_random_sample_visits():
    files = [f"img_{i:03d}.jpg" for i in range(37)]
    random.seed(11)
    want = [[files.index(f) for f in random.sample(files, 9)] for _ in range(5)]
    got = diversity_oracle.draw_samples(37, 5, 9, seed=11)
    assert got.tolist() == want


def test_driver_host_logic_with_oracle_backend(tmp_path, monkeypatch):
    def fake(emb, samples, first=0):
        return torch.tensor(diversity_oracle.diversity_order(emb.numpy(), samples, first), dtype=torch.int32)
    monkeypatch.setattr(diversity_driver, "diversity_order_indices", fake)
    n, d = 30, 12
    emb = _unit(n, d, 5)
    files = [str(tmp_path / f"im{i:02d}.jpg") for i in range(n)]
    for i, f in enumerate(files):
        t = torch.from_numpy(emb[i]).unsqueeze(0)
        if i % 2:                                                      # both .pt layouts
            torch.save({"square_padded_crop": t, "centre_crop": t * 0}, f.replace(".jpg", ".pt"))
        else:
            torch.save({"ViT-L-14/openai": {"square_padded_crop": t, "centre_crop": t * 0}}, f.replace(".jpg", ".pt"))
    random.seed(2)
    got = diversity_driver.diversity_ordered_image_files(files, str(tmp_path), total_n_ordered_imgs=8, sample_size=6, device="cpu")
    want_order = diversity_oracle.diversity_order(emb, diversity_oracle.draw_samples(n, 8, 6, seed=2), first=0)
    head = [files[0]] + [files[i] for i in want_order]
    assert got[:9] == head
    assert got[9:] == [f for f in files if f not in set(head)] and sorted(got) == sorted(files)
    # fewer files than requested steps: len - 1 steps (:146); a sample larger than the population raises as random.sample does
    random.seed(2)
    assert len(diversity_driver.diversity_ordered_image_files(files[:7], str(tmp_path), 500, 5, device="cpu")) == 7
    with pytest.raises(ValueError):
        diversity_driver.diversity_ordered_image_files(files[:4], str(tmp_path), 500, 100, device="cpu")
    assert diversity_driver.diversity_ordered_image_files(files[:1], str(tmp_path), 500, 100, device="cpu") == files[:1]


def test_product_entry_refuses_cpu_tensors():
    with pytest.raises(Exception):
        diversity_driver.diversity_order_indices(torch.zeros(4, 8), np.zeros((1, 2), np.int32))
