"""TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/jpeg_ref.cpp: the JPEG decoding arithmetic of the HIP kernels run on the
CPU, so that it can be compared with Pillow (the decoder behind /root/reference/utils/embedder.py:167) without a GPU."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REASONS = ["ok", "not a JPEG file", "progressive", "precision", "components", "sampling", "multi scan", "colour space", "tables",
           "arithmetic", "too large", "truncated", "corrupt"]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libjpeg_ref.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(path)
        _LIB.jpeg_ref_info.restype = ctypes.c_int
        _LIB.jpeg_ref_decode.restype = ctypes.c_int
        _LIB.jpeg_ref_decode_parallel.restype = ctypes.c_int
    return _LIB


def info(data: bytes):
    """(reason code, width, height, components)"""
    w, h, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = _lib().jpeg_ref_info(data, ctypes.c_size_t(len(data)), ctypes.byref(w), ctypes.byref(h), ctypes.byref(n))
    return rc, w.value, h.value, n.value


def decode(data: bytes):
    """uint8 [H, W, 3] or raises ValueError(reason)"""
    rc, w, h, _ = info(data)
    if rc:
        raise ValueError(REASONS[rc] if rc < len(REASONS) else str(rc))
    out = np.empty((h, w, 3), dtype=np.uint8)
    rc = _lib().jpeg_ref_decode(data, ctypes.c_size_t(len(data)), out.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise ValueError(f"decode failed: {rc}")
    return out


def decode_parallel(data: bytes, sub_bytes: int = 128, max_passes: int = 64):
    """(uint8 [H, W, 3], synchronisation passes) through the PARALLEL entropy decoder, its threads run one after the other"""
    rc, w, h, _ = info(data)
    if rc:
        raise ValueError(REASONS[rc] if rc < len(REASONS) else str(rc))
    out = np.empty((h, w, 3), dtype=np.uint8)
    passes = ctypes.c_int()
    rc = _lib().jpeg_ref_decode_parallel(data, ctypes.c_size_t(len(data)), out.ctypes.data_as(ctypes.c_void_p), int(sub_bytes), int(max_passes),
                                         ctypes.byref(passes))
    if rc:
        raise ValueError(f"decode failed: {rc}")
    return out, passes.value
