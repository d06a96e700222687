"""TEST INFRASTRUCTURE ONLY — CPU restatement of the near-duplicate search.

Follows /root/reference/_2_remove_duplicates.py:63-80: stack fp16 embeddings (:38 casts each to
float16), normalise rows (:67), similarity = E_hat @ E_hat.T (:69), pairs = where(triu(S, 1) > thr)
in row-major (i < j) order (:74-76), values = S[i, j] (:80).
Pinned against the reference itself: tests/golden/make_golden.py runs the reference's
`find_near_duplicates` on a planted-pair set and stores the pairs it reports.
"""
from __future__ import annotations

import torch


@torch.no_grad()
def near_duplicates(emb_fp16: torch.Tensor, threshold: float, block: int = 2048):
    """Returns (pairs int64 [P,2] sorted row-major, values fp16 [P]) exactly as the reference would."""
    e = emb_fp16.to(torch.float16)
    e = e / torch.norm(e, dim=1, keepdim=True)
    n = e.shape[0]
    pairs, vals = [], []
    for i0 in range(0, n, block):            # blocked only to bound memory; same arithmetic per entry
        s = torch.matmul(e[i0:i0 + block], e.T)
        ii, jj = torch.where(torch.triu(s, diagonal=1 + i0) > threshold)
        pairs.append(torch.stack([ii + i0, jj], 1))
        vals.append(s[ii, jj])
    return torch.cat(pairs, 0), torch.cat(vals, 0)


@torch.no_grad()
def similarity_fp32(emb_fp16: torch.Tensor) -> torch.Tensor:
    """fp32 recomputation used for the tolerance band around the fp16 threshold."""
    e = emb_fp16.to(torch.float16)
    e = (e / torch.norm(e, dim=1, keepdim=True)).float()
    return e @ e.T
