"""TEST INFRASTRUCTURE ONLY — CPU restatement of the regressor training of /root/reference/_4_train_model.py.

Model (utils/nn_model.py:6-41): Linear -> LeakyReLU(0.01) -> Dropout(p) per hidden layer, Linear -> Sigmoid at the end.
One optimisation step (:201-207): MSELoss (mean over the batch, outputs squeezed, :204), backward, torch.optim.Adam with
`weight_decay` added to the gradient (L2, not AdamW), betas (0.9, 0.999), eps 1e-8, bias-corrected as torch does
(`denom = sqrt(v) / sqrt(1 - b2^t) + eps`, `p -= lr / (1 - b1^t) * m / denom`).  The learning rate follows
CosineAnnealingWarmRestarts(T_0 = restart_epochs, T_mult = 1, eta_min = min_lr), stepped once per epoch (:210).
Dropout masks are the one thing that cannot be shared with torch (its generator is not reproducible outside torch):
they come from the counter-based hash below, which the HIP kernels evaluate identically, and the comparison with the
reference's own classes (tests/golden/make_golden.py: the reference SimpleFC + torch Adam + the torch scheduler) is made
at dropout_prob = 0.  Pinned by tests/golden/train_small.npz.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

NEG_SLOPE = 0.01
B1, B2, EPS = 0.9, 0.999, 1e-8


def lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep(seed: int, step: int, layer: int, rows: int, cols: int, p: float) -> np.ndarray:
    """keep[r][c] of hidden layer `layer` (0-based) at optimisation step `step`: hash(seed, step, layer, r, c) >= p * 2^32"""
    if p <= 0.0:
        return np.ones((rows, cols), dtype=bool)
    r = np.arange(rows, dtype=np.uint32)[:, None]
    c = np.arange(cols, dtype=np.uint32)[None, :]
    with np.errstate(over="ignore"):
        h = lowbias32(np.uint32(seed) ^ lowbias32(np.uint32(step) + np.uint32(0x9e3779b9) * np.uint32(layer + 1)))
        h = lowbias32(h ^ lowbias32(r * np.uint32(0x85ebca6b) + c + np.uint32(1)))
    thr = np.uint32(min(int(p * 4294967296.0), 4294967295))
    return h >= thr


def cosine_lr(base_lr: float, eta_min: float, T_0: int, epoch: int) -> float:
    """learning rate used DURING epoch `epoch` (0-based): the scheduler has been stepped `epoch` times"""
    t = epoch % T_0
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / T_0)) / 2.0


class Trainer:
    def __init__(self, Ws: Sequence[np.ndarray], bs: Sequence[np.ndarray], weight_decay: float, dropout_p: float = 0.0, seed: int = 0,
                 dtype=np.float32):
        self.dt = dtype
        self.W = [np.array(w, dtype=dtype) for w in Ws]
        self.b = [np.array(b, dtype=dtype) for b in bs]
        self.mW = [np.zeros_like(w) for w in self.W]; self.vW = [np.zeros_like(w) for w in self.W]
        self.mb = [np.zeros_like(b) for b in self.b]; self.vb = [np.zeros_like(b) for b in self.b]
        self.wd, self.p, self.seed, self.t = weight_decay, dropout_p, seed, 0

    def forward(self, x, train: bool, step: int = 0):
        a, zs, acts, keeps = x.astype(self.dt), [], [x.astype(self.dt)], []
        L = len(self.W)
        for l in range(L):
            z = a @ self.W[l].T + self.b[l]
            zs.append(z)
            if l < L - 1:
                a = np.where(z > 0, z, self.dt(NEG_SLOPE) * z)
                if train and self.p > 0:
                    k = dropout_keep(self.seed, step, l, z.shape[0], z.shape[1], self.p)
                    a = np.where(k, a / self.dt(1.0 - self.p), self.dt(0))
                    keeps.append(k)
                else:
                    keeps.append(None)
            else:
                a = 1.0 / (1.0 + np.exp(-z))
            acts.append(a.astype(self.dt))
        return acts, zs, keeps

    def step(self, x, t, lr: float) -> float:
        """one Adam step on the batch; returns the batch MSE (before the update)"""
        B = x.shape[0]
        acts, zs, keeps = self.forward(x, True, self.t)
        y = acts[-1][:, 0]
        tt = t.astype(self.dt)
        loss = float(np.mean((y.astype(np.float64) - tt) ** 2))
        dz = ((2.0 / B) * (y - tt) * y * (1 - y)).astype(self.dt)[:, None]
        gW, gb = [None] * len(self.W), [None] * len(self.W)
        for l in range(len(self.W) - 1, -1, -1):
            gW[l] = dz.T @ acts[l]
            gb[l] = dz.sum(0)
            if l > 0:
                da = dz @ self.W[l]
                if keeps[l - 1] is not None:
                    da = np.where(keeps[l - 1], da / self.dt(1.0 - self.p), self.dt(0))
                dz = (da * np.where(zs[l - 1] > 0, self.dt(1), self.dt(NEG_SLOPE))).astype(self.dt)
        self.t += 1
        c1, c2 = 1.0 - B1 ** self.t, 1.0 - B2 ** self.t
        for P, G, M, V in ((self.W, gW, self.mW, self.vW), (self.b, gb, self.mb, self.vb)):
            for l in range(len(P)):
                g = (G[l] + self.dt(self.wd) * P[l]).astype(self.dt)
                M[l] = (self.dt(B1) * M[l] + self.dt(1 - B1) * g).astype(self.dt)
                V[l] = (self.dt(B2) * V[l] + self.dt(1 - B2) * g * g).astype(self.dt)
                denom = np.sqrt(V[l]) / self.dt(math.sqrt(c2)) + self.dt(EPS)
                P[l] = (P[l] - self.dt(lr / c1) * M[l] / denom).astype(self.dt)
        return loss

    def epoch(self, X, T, order: np.ndarray, batch_size: int, lr: float) -> float:
        losses = []
        for b0 in range(0, len(order), batch_size):
            idx = order[b0:b0 + batch_size]
            losses.append(self.step(X[idx], T[idx], lr))
        return float(np.mean(losses))

    def evaluate(self, X, T, batch_size: int) -> float:
        """mean over batches of the batch MSE in eval mode (:132-166)"""
        losses = []
        for b0 in range(0, len(X), batch_size):
            y = self.forward(X[b0:b0 + batch_size], False)[0][-1][:, 0]
            losses.append(float(np.mean((y.astype(np.float64) - T[b0:b0 + batch_size]) ** 2)))
        return float(np.mean(losses)) if losses else -1.0
