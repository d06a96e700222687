"""Host logic of the near-duplicate search (/root/reference/_2_remove_duplicates.py:63-80 -> dedup_find_pairs): the execution
order of the upper-triangular tile list (`dedup_tile_order`, no device work) must hold every tile tn >= tm exactly once for
any grid, and must keep the tiles that share an XCD in a round inside a few operand panels."""
import ctypes

import numpy as np
import pytest

from clip_assisted_data_labeling_amd import _lib


def _order(tt, grid):
    lib = _lib.load()
    n = tt * (tt + 1) // 2
    buf = (ctypes.c_uint * n)()
    _lib.check(lib.dedup_tile_order(tt, grid, buf, n), "dedup_tile_order")
    return np.frombuffer(buf, dtype=np.uint32).copy()


@pytest.mark.parametrize("tt", [1, 2, 3, 7, 8, 9, 33, 40, 391])
@pytest.mark.parametrize("grid", [1, 5, 8, 64, 256, 304])
def test_tile_order_is_a_permutation_of_the_upper_triangle(tt, grid):
    n = tt * (tt + 1) // 2
    o = _order(tt, min(grid, n))
    tm, tn = (o & 0xFFFF).astype(np.int64), (o >> 16).astype(np.int64)
    assert (tn >= tm).all() and tn.max() == tt - 1 and tm.min() == 0
    assert len(np.unique(tm * 65536 + tn)) == n == len(o)


def test_tile_order_keeps_an_xcds_tiles_in_few_panels():
    """BASELINE.json configs[4]: 100 000 rows = 391 tiles per side on 256 workgroups.  Workgroups b, b + 8, ... share an XCD
    (round-robin dispatch); per round each XCD should touch ~12 panels (8 x 4 block), not the 33 of a row-major walk."""
    tt, G = 391, 256
    o = _order(tt, G)
    rounds = len(o) // G
    panels = []
    for r in range(rounds):
        blk = o[r * G:(r + 1) * G]
        for x in range(8):
            w = blk[x::8]
            panels.append(len(set((w & 0xFFFF).tolist()) | set(((w >> 16) + (1 << 20)).tolist())))
    assert np.mean(panels) < 12.5 and max(panels) <= 24, (np.mean(panels), max(panels))
    # consecutive rounds of one XCD stay in one super-row most of the time (its 8 A panels are re-used)
    same = 0
    for r in range(rounds - 1):
        a = set((o[r * G:(r + 1) * G][0::8] & 0xFFFF).tolist())
        b = set((o[(r + 1) * G:(r + 2) * G][0::8] & 0xFFFF).tolist())
        same += a == b
    assert same / (rounds - 1) > 0.8


def test_tile_order_rejects_bad_arguments():
    lib = _lib.load()
    buf = (ctypes.c_uint * 4)()
    assert lib.dedup_tile_order(3, 8, buf, 4) != 0            # 6 tiles do not fit 4 entries
    assert lib.dedup_tile_order(0, 8, buf, 4) != 0
    assert lib.dedup_tile_order(70000, 8, buf, 4) != 0
