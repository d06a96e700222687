"""TEST INFRASTRUCTURE ONLY — CPU restatement of the similarity search of /root/reference/tools/find_similar_imgs.py.

* context embedding = mean over the context set of each file's concatenated [model][crop] embedding (:28-62);
* distance (`compute_distance`, :88-94): "l2" = torch.nn.functional.pairwise_distance(q, e, p=2, eps=1e-6)
  = || q - e + 1e-6 ||_2 (the eps is added to every component of the difference); "cosine" =
  (1 - cosine_similarity(q, e, dim=-1)) / 2 with torch's eps = 1e-8 clamp on each norm;
* top-N (`topN`, :67-85): keep the N smallest distances seen; a candidate only replaces the current worst when it is
  strictly smaller, so among equal distances the earlier file stays.  The reference reports them in slot order; this
  restatement (and the HIP path) reports them ascending by (distance, index) — the same SET whenever distances differ.
Pinned against the reference's own `compute_distance` and `topN` by tests/golden/make_golden.py (simsearch_small.npz).
"""
from __future__ import annotations

import numpy as np


def distances(emb: np.ndarray, query: np.ndarray, measure: str) -> np.ndarray:
    e = emb.astype(np.float32)
    q = query.astype(np.float32)
    if measure == "l2":
        diff = (q[None, :] - e) + np.float32(1e-6)
        return np.sqrt((diff.astype(np.float64) ** 2).sum(-1)).astype(np.float32)
    if measure == "cosine":
        dot = (e.astype(np.float64) * q.astype(np.float64)[None, :]).sum(-1)
        ne = np.maximum(np.sqrt((e.astype(np.float64) ** 2).sum(-1)), 1e-8)
        nq = max(float(np.sqrt((q.astype(np.float64) ** 2).sum())), 1e-8)
        return ((1.0 - dot / (ne * nq)) / 2.0).astype(np.float32)
    raise NotImplementedError(f"Similarity measure {measure} not implemented!")


def top_n(dist: np.ndarray, n: int):
    """indices and values of the n smallest distances, ascending, ties by lower index; a NaN counts as +inf."""
    d = np.where(np.isnan(dist), np.float32(np.inf), dist).astype(np.float32)
    order = np.lexsort((np.arange(len(d)), d))[: min(n, len(d))]
    return order.astype(np.int64), d[order]
