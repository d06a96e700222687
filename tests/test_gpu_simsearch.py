"""Similarity search on the GPU (simsearch_distances + simsearch_topn) against the reference's own vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import similar_driver
from oracle import simsearch_oracle

pytestmark = pytest.mark.gpu
DIST_TOL = 2e-6          # fp32 sums of <= 3072 terms of O(1) against float64


@pytest.mark.parametrize("measure", ["l2", "cosine"])
def test_matches_reference_golden(gpu, golden_dir, measure):
    g = np.load(os.path.join(golden_dir, "simsearch_small.npz"))
    emb, q, top = torch.from_numpy(g["emb"]), g["query"], int(g["top_n"])
    idx, val = similar_driver.nearest(emb, q, measure, top, gpu)
    assert np.abs(val - g[f"dist_{measure}"][idx]).max() <= DIST_TOL
    assert sorted(idx.tolist()) == g[f"kept_{measure}"].tolist()              # the set the reference's topN keeps
    assert np.all(np.diff(val) >= 0)
    if measure == "l2":
        i3, i7 = idx.tolist().index(3) if 3 in idx else None, idx.tolist().index(7) if 7 in idx else None
        assert (i3 is None) == (i7 is None) and (i3 is None or i3 < i7)      # the planted tie: lower index first
    # all distances: ask for every row
    idx_all, val_all = similar_driver.nearest(emb, q, measure, emb.shape[0], gpu)
    assert sorted(idx_all.tolist()) == list(range(emb.shape[0]))
    assert np.abs(val_all[np.argsort(idx_all)] - g[f"dist_{measure}"]).max() <= DIST_TOL


@pytest.mark.parametrize("n,d,dtype", [(1, 8, torch.float32), (5, 7, torch.float32), (3000, 768, torch.float16),
                                       (200_000, 96, torch.float32), (1_000_000, 64, torch.float16)])
def test_large_sets_odd_widths_and_fp16_vs_oracle(gpu, n, d, dtype):
    g = torch.Generator().manual_seed(n + d)
    emb = torch.randn(n, d, generator=g).to(dtype)
    q = torch.randn(d, generator=g).numpy()
    for measure in ("l2", "cosine"):
        ref = simsearch_oracle.distances(emb.float().numpy(), q, measure)
        top = min(n, 37)
        idx, val = similar_driver.nearest(emb, q, measure, top, gpu)
        assert len(idx) == top and len(set(idx.tolist())) == top
        assert np.abs(val - ref[idx]).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))
        # the selected set is the oracle's up to distances closer than the tolerance
        oi, ov = simsearch_oracle.top_n(ref, top)
        assert val[-1] <= ov[-1] + 1e-4 and np.all(np.diff(val) >= 0)
        assert len(set(idx.tolist()) & set(oi.tolist())) >= top - 2


def test_strided_rows_select_one_crop_of_a_packed_block(gpu):
    g = torch.Generator().manual_seed(1)
    block = torch.randn(500, 4, 32, generator=g)                               # [n][crops][E] as the packed store keeps it
    q = torch.randn(32, generator=g).numpy()
    for crop in range(4):
        idx, val = similar_driver.nearest(block, q, "l2", 9, gpu, row_offset=crop * 32, row_stride=128, d=32)
        oi, ov = simsearch_oracle.top_n(simsearch_oracle.distances(block[:, crop].numpy(), q, "l2"), 9)
        assert idx.tolist() == oi.tolist() and np.abs(val - ov).max() <= DIST_TOL


def test_topn_edge_cases(gpu):
    d = torch.tensor([[2.0], [1.0], [float("nan")], [1.0], [0.5]])            # 1-d "embeddings": l2 to q = 0 is |x + 1e-6|
    idx, val = similar_driver.nearest(d, np.zeros(1, np.float32), "l2", 4, gpu)
    assert idx.tolist() == [4, 1, 3, 0]
    idx, val = similar_driver.nearest(d, np.zeros(1, np.float32), "l2", 99, gpu)   # top_n > n: clipped, NaN last as +inf
    assert idx.tolist() == [4, 1, 3, 0, 2] and np.isinf(val[-1])
    same = torch.ones(5000, 4)
    idx, _ = similar_driver.nearest(same, np.ones(4, np.float32), "cosine", 6, gpu)
    assert idx.tolist() == [0, 1, 2, 3, 4, 5]                                  # all equal: the first rows win, in order
    assert similar_driver.nearest(torch.zeros(0, 4), np.ones(4, np.float32), "l2", 3, gpu)[0].size == 0
    with pytest.raises(NotImplementedError):
        similar_driver.nearest(same, np.ones(4, np.float32), "dot", 3, gpu)
