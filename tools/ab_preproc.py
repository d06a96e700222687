#!/usr/bin/env python3
"""Developer: time of the crop front end (preproc_crops_u8_batch: tables + horizontal + vertical pass) alone, on decoded images in HBM.
    CLIPENC_LIB_PATH=<variant .so> python tools/ab_preproc.py [--n 512] [--size 512]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_assisted_data_labeling_amd.preprocess import GpuCropper

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--size", type=int, nargs="+", default=[512, 1536])
a = ap.parse_args()
cropper = GpuCropper(224, "cuda:0")
for size in a.size:
    rs = np.random.RandomState(0)
    n = a.n if size <= 1024 else max(a.n // 8, 8)
    imgs = [torch.from_numpy(rs.randint(0, 256, (size, size + 16 * (i % 5), 3), dtype=np.uint8)).cuda() for i in range(n)]
    out, _ = cropper.batch(imgs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        e0.record(); out, _ = cropper.batch(imgs); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{os.path.basename(os.environ.get('CLIPENC_LIB_PATH', 'shipped')):28s} {n} images of ~{size}^2: {min(ts):.3f} ms per call (min of 5), "
          f"checksum {int(out.sum(dtype=torch.int64))}", flush=True)
