"""TEST INFRASTRUCTURE ONLY — ctypes front end of oracle/fcreg_oracle.c plus a numpy restatement.

Both follow /root/reference/utils/nn_model.py:21-41; see fcreg_oracle.c for the pinning note.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libfcreg_oracle.so")


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def _lib():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.fcreg_oracle_forward.restype = ctypes.c_int
    return lib


def forward_c(weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], x: np.ndarray,
              negative_slope: float = 0.01) -> np.ndarray:
    lib = _lib()
    n = len(weights)
    sizes = [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]
    Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty((x.shape[0], sizes[-1]), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    Wp = (fp * n)(*[w.ctypes.data_as(fp) for w in Ws])
    bp = (fp * n)(*[b.ctypes.data_as(fp) for b in bs])
    rc = lib.fcreg_oracle_forward(n, (ctypes.c_int * (n + 1))(*sizes), Wp, bp,
                                  ctypes.c_float(negative_slope), x.ctypes.data_as(fp),
                                  x.shape[0], y.ctypes.data_as(fp))
    if rc != 0:
        raise RuntimeError("fcreg_oracle_forward failed")
    return y


def forward_np(weights, biases, x, negative_slope: float = 0.01) -> np.ndarray:
    h = np.asarray(x, dtype=np.float32)
    n = len(weights)
    for l, (w, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(w, np.float32).T + np.asarray(b, np.float32)
        if l < n - 1:
            h = np.where(h >= 0, h, negative_slope * h).astype(np.float32)
        else:
            h = (1.0 / (1.0 + np.exp(-h))).astype(np.float32)
    return h
