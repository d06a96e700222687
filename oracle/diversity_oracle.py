"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the diversity ordering of a labeling
session, /root/reference/_3_label_images.py:128-177.

Parity unpinned at the reference boundary: `_3_label_images.py` imports cv2, natsort and tkinter, none of which exist in
the build container, so the function cannot be run here and the reference holds no test or fixture for it.  The
restatement uses the same torch CPU ops in the same order (`cosine_similarity_matrix` :129-133: rows divided by their
norms, one matmul; :161-167: column maxima over the chosen set, `torch.argmin`), with the per-step samples passed in as
indices: `random.sample(image_files, sample_size)` (:148) picks POSITIONS that depend only on (len, k) and the RNG
state, so `random.sample(range(n), k)` under the same seed reproduces them.
"""
import random
from typing import List, Sequence

import numpy as np
import torch


def draw_samples(n: int, steps: int, sample_size: int, seed=None) -> np.ndarray:
    """The index sets the reference's `random.sample(image_files, sample_size)` visits, one row per step (:146-148)."""
    if seed is not None:
        random.seed(seed)
    return np.array([random.sample(range(n), sample_size) for _ in range(steps)], dtype=np.int32).reshape(steps, sample_size)


def cosine_similarity_matrix(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a_norm = a / a.norm(dim=1, keepdim=True)            # :130
    b_norm = b / b.norm(dim=1, keepdim=True)            # :131
    return torch.matmul(a_norm, b_norm.t())             # :132


def diversity_order(emb: np.ndarray, samples: Sequence[Sequence[int]], first: int = 0) -> List[int]:
    """emb [n][E]; returns the index appended at every step (the reference's img_files[1:], :169)."""
    e = torch.from_numpy(np.ascontiguousarray(emb, dtype=np.float32))
    chosen = e[first].unsqueeze(0)                       # :141-143
    order = []
    with torch.no_grad():
        for s in samples:                                # :146
            idx = torch.as_tensor(np.asarray(s, dtype=np.int64))
            sample_embeddings = e[idx]                   # :154-155
            similarities = cosine_similarity_matrix(chosen, sample_embeddings)   # :158
            max_val, _ = torch.max(similarities, dim=0)                          # :161
            index_of_min = int(torch.argmin(max_val).item())                     # :164
            order.append(int(idx[index_of_min]))                                 # :167
            chosen = torch.cat((chosen, sample_embeddings[index_of_min].unsqueeze(0)), dim=0)   # :168-171
    return order


def margins(emb: np.ndarray, samples, first: int = 0) -> np.ndarray:
    """Gap between the smallest and second smallest column maximum at every step (float64): a step whose gap is below the
    fp32 accumulation-order noise (a few 1e-7) has no single right answer."""
    e = np.asarray(emb, dtype=np.float64)
    e = e / np.linalg.norm(e, axis=1, keepdims=True)
    order = diversity_order(emb, samples, first)
    ms = np.full(e.shape[0], -np.inf)
    cur, out = first, []
    for t, s in enumerate(samples):
        ms = np.maximum(ms, e @ e[cur])
        v = np.sort(ms[np.asarray(s)])
        out.append(v[1] - v[0] if len(v) > 1 else np.inf)
        cur = order[t]
    return np.array(out)
