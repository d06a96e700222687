"""Store-scale SimpleFC path (fcreg_mfma_kernel: fp32 matrix pipe, 128 rows per workgroup, activations in registers) against
the C oracle (/root/reference/utils/nn_model.py:21-41 restated) and against the small-batch kernel: same `fcreg_forward`
entry, selected by the row count.  Tolerance: scores within 1e-4 (north_star); both kernels are exact-fp32 fma chains, so
they agree to a few ulp."""
import ctypes
import os

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib
from clip_assisted_data_labeling_amd.nn_model import HipRegressor
from oracle import fcreg_oracle
from tests.helpers import np_fc_weights

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4


def _unit_rows(n, d, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return (x / x.norm(dim=-1, keepdim=True)).contiguous()


@pytest.mark.parametrize("sizes,n_rows,slope", [([3072, 264, 128, 64, 1], 5000, 0.01),      # 4-crop regressor; 5000 = 39 x 128 + 8 rows
                                                 ([768, 264, 128, 64, 1], 4096, 0.01),       # shipped checkpoint shape
                                                 ([1536, 288, 100, 3], 4321, 0.2),           # 3 layers, 3 outputs, widths not multiples of 32
                                                 ([64, 7], 4100, 0.01)])                     # single layer
def test_store_scale_kernel_matches_oracle(gpu, sizes, n_rows, slope):
    Ws, bs = np_fc_weights(sizes, 11)
    Ws = [w * 3.0 for w in Ws]                                         # spread the scores over (0, 1)
    reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], slope, gpu)
    x = _unit_rows(n_rows, sizes[0], 5, gpu) * 4.0
    y = reg(x).cpu().numpy()
    ref = fcreg_oracle.forward_c(Ws, bs, x.cpu().numpy(), slope)
    assert y.shape == ref.shape and np.isfinite(y).all()
    assert np.abs(y - ref).max() < SCORE_TOL
    assert ref.std() > 0.01
    # the same rows through the small-batch kernel (row counts below the switch-over) give the same scores
    y_small = torch.cat([reg(x[i:i + 1000]) for i in range(0, 3000, 1000)]).cpu().numpy()
    assert np.abs(y_small - y[:3000]).max() < 2e-6
    reg.close()


def test_store_scale_kernel_shipped_checkpoint_and_crop_segments(gpu, golden_dir):
    """The reference's shipped regressor on 8 192 stored rows, read in place from a packed [n][4 crops][768] block with the
    crop selection of model.crop_names (_5_predict_labels.py:79), and a 2-crop selection in non-storage order."""
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "regressor_shipped.npz"))
    n = int(g["n_layers"])
    Ws, bs = [g[f"W{i}"] for i in range(n)], [g[f"b{i}"] for i in range(n)]
    reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], float(g["negative_slope"]), gpu)
    assert np.abs(reg(torch.from_numpy(g["x"]).to(gpu)).cpu().numpy() - g["y"]).max() < SCORE_TOL      # golden rows (small kernel)
    N, E = 8192, 768
    store = _unit_rows(N * 4, E, 9, gpu).view(N, 4, E)
    y = torch.empty((N, 1), device=gpu)
    off = (ctypes.c_int * 1)(0 * E)                                    # centre_crop
    _lib.check(lib.fcreg_forward(reg.handle, store.data_ptr(), N, 4 * E, 1, E, off, y.data_ptr(), _lib.current_stream_ptr(gpu)), "fcreg")
    torch.cuda.synchronize()
    ref = fcreg_oracle.forward_c(Ws, bs, store[:, 0, :].cpu().numpy(), float(g["negative_slope"]))
    assert np.abs(y.cpu().numpy() - ref).max() < SCORE_TOL
    Ws2, bs2 = np_fc_weights([2 * E, 264, 128, 64, 1], 4)
    reg2 = HipRegressor([torch.from_numpy(w) for w in Ws2], [torch.from_numpy(b) for b in bs2], 0.01, gpu)
    off2 = (ctypes.c_int * 2)(3 * E, 1 * E)                            # subcrop2, square_padded_crop
    _lib.check(lib.fcreg_forward(reg2.handle, store.data_ptr(), N, 4 * E, 2, E, off2, y.data_ptr(), _lib.current_stream_ptr(gpu)), "fcreg")
    torch.cuda.synchronize()
    feats = torch.cat([store[:, 3], store[:, 1]], dim=1).cpu().numpy()
    assert np.abs(y.cpu().numpy() - fcreg_oracle.forward_c(Ws2, bs2, feats)).max() < SCORE_TOL
    reg.close(); reg2.close()
