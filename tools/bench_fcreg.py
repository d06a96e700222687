#!/usr/bin/env python3
"""Developer timing of the stand-alone regressor at store scale: N rows x 3072 fp32 (4 crops x 768) -> score, the arithmetic of
/root/reference/_5_predict_labels.py:133-135 over a whole packed store that is already in HBM."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.nn_model import HipRegressor
from tools.quick_bench import timeit
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=1 << 20); ap.add_argument("--crops", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda", 0)
sizes = [args.crops * 768, 264, 128, 64, 1]
rs = np.random.RandomState(0)
Ws = [rs.uniform(-1, 1, (sizes[i + 1], sizes[i])).astype(np.float32) / np.sqrt(sizes[i]) for i in range(4)]
bs = [rs.uniform(-0.1, 0.1, (sizes[i + 1],)).astype(np.float32) for i in range(4)]
reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], 0.01, dev)
x = torch.randn(args.rows, sizes[0], device=dev)
ms = timeit(lambda: reg(x), iters=5, warmup=2)
gb = args.rows * sizes[0] * 4 / 1e9
fl = 2.0 * args.rows * sum(sizes[i] * sizes[i + 1] for i in range(4))
print(f"fcreg {args.rows} rows x {sizes[0]}: {ms:.2f} ms  {gb / ms * 1e3:.0f} GB/s of input  {fl / ms / 1e9:.1f} TFLOP/s fp32 "
      f"({fl / ms / 1e9 / 157.3 * 100:.0f}% of the 157.3 TFLOP/s fp32 matrix peak)  {args.rows / ms * 1e3 / 1e6:.1f} M rows/s")
for rows in (512, 2048):
    xs = x[:rows].contiguous()
    print(f"fcreg {rows} rows (small-batch kernel): {timeit(lambda: reg(xs), iters=20, warmup=3):.3f} ms")
