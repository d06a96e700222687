"""The GPU front end's resampling tables (computed on the host inside libclipenc_hip.so) and its two integer passes,
pinned against Pillow itself on the CPU: bit-exact for up- and down-scaling, odd sizes and the crop geometry."""
import ctypes

import numpy as np
import pytest
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import _lib
from clip_assisted_data_labeling_amd.preprocess import ClipValTransform, crop_box_table, extract_crops

PB = 22


def axis_tables(in_size, out_size, out0=0, n_out=None):
    lib = _lib.load()
    n_out = out_size - out0 if n_out is None else n_out
    cap = n_out * (int(np.ceil(2.0 * max(in_size / out_size, 1.0))) * 2 + 1)
    bounds = (ctypes.c_int * (n_out * 2))()
    kk = (ctypes.c_int * cap)()
    ks = ctypes.c_int()
    _lib.check(lib.preproc_axis_tables(in_size, out_size, out0, n_out, bounds, kk, cap, ctypes.byref(ks)), "tables")
    return np.array(bounds).reshape(n_out, 2), np.array(kk).reshape(n_out, ks.value), ks.value


def resample_axis(img, out_size, axis):
    """One Pillow pass with the library's tables: int32 accumulate from 1 << 21, arithmetic >> 22, clip to uint8."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk, _ = axis_tables(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for i in range(out_size):
        x0, n = bounds[i]
        acc = (1 << (PB - 1)) + np.tensordot(kk[i, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[i] = np.clip(acc >> PB, 0, 255)
    return np.moveaxis(out, 0, axis)


@pytest.mark.parametrize("w,h,nw,nh", [(500, 300, 373, 224), (97, 333, 224, 768), (224, 224, 224, 224), (1500, 1000, 336, 224),
                                       (64, 48, 298, 224), (301, 299, 225, 224)])
def test_tables_and_integer_passes_equal_pillow_bicubic(w, h, nw, nh):
    rs = np.random.RandomState(w * 7 + h)
    arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(arr).resize((nw, nh), Image.BICUBIC))
    got = resample_axis(resample_axis(arr, nw, 1), nh, 0)           # Pillow: horizontal pass first, then vertical
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), int(np.abs(got.astype(int) - ref.astype(int)).max())


def test_partial_tables_are_slices_of_full_tables():
    b_full, k_full, ks = axis_tables(1000, 336)
    b_part, k_part, ks2 = axis_tables(1000, 336, 56, 224)
    assert ks == ks2 and np.array_equal(b_part, b_full[56:280]) and np.array_equal(k_part, k_full[56:280])
    assert (k_full.sum(1) > (1 << PB) - 64).all() and (k_full.sum(1) < (1 << PB) + 64).all()   # weights sum to ~1.0


def test_emulated_front_end_equals_pillow_transform():
    """Whole front end (crop/pad geometry -> resize -> centre crop) emulated with the library tables == the PIL path."""
    rs = np.random.RandomState(4)
    R = 224
    for (w, h) in [(640, 427), (300, 500), (224, 224), (1000, 60)]:
        arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        img = Image.fromarray(arr)
        crops, names = extract_crops(img)
        want = [ClipValTransform(R).to_uint8(c).numpy() for c in crops]
        rows, names2 = crop_box_table(w, h)
        assert names == names2
        for (kind, a, b, c, d), ref in zip(rows, want):
            if kind == 0:
                canvas = arr[b:d, a:c]
            else:
                canvas = np.zeros((a, a, 3), np.uint8)
                canvas[c:c + h, b:b + w] = arr
            ch, cw = canvas.shape[:2]
            nw, nh = (R, int(R * ch / cw)) if cw <= ch else (int(R * cw / ch), R)
            res = resample_axis(resample_axis(canvas, nw, 1), nh, 0) if (nw, nh) != (cw, ch) else canvas
            top, left = int(round((nh - R) / 2.0)), int(round((nw - R) / 2.0))
            got = res[top:top + R, left:left + R].transpose(2, 0, 1)
            assert np.array_equal(got, ref), (w, h, kind)
