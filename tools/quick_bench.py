#!/usr/bin/env python3
"""Developer timing of the individual kernels and the ViT-L/14 chain (not the judged bench: see bench.py)."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib, vit_config  # noqa: E402
from clip_assisted_data_labeling_amd.embedder import HipViT  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=256)
    ap.add_argument("--skip-encode", action="store_true")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = _lib.current_stream_ptr(dev)
    T = args.crops * 257
    print(f"device {torch.cuda.get_device_name(0)}; T = {T} token rows")
    for (n, k, name) in [(3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]:
        a = torch.randn(T, k, device=dev).to(torch.bfloat16)
        w = torch.randn(n, k, device=dev).to(torch.bfloat16)
        o = torch.empty(T, n, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: lib.clipenc_op_gemm_nt(a.data_ptr(), w.data_ptr(), T, n, k, 0, 1, None, o.data_ptr(), st))
        print(f"gemm {name:4s} M={T} N={n} K={k}: {ms:8.3f} ms  {2.0 * T * n * k / ms / 1e9:8.1f} TFLOP/s")
        del a, w, o
    qkv = torch.randn(T, 3072, device=dev).to(torch.bfloat16)
    o = torch.empty(T, 1024, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: lib.clipenc_op_attention(qkv.data_ptr(), o.data_ptr(), args.crops, 257, 1024, 16, st))
    fl = 4.0 * args.crops * 16 * 257 * 257 * 64
    print(f"attention {args.crops} crops: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  {(T * 4096 * 2) / ms / 1e6:8.1f} GB/s")
    del qkv, o
    if args.skip_encode:
        return
    cfg = vit_config.ARCHS["ViT-L-14"]
    t0 = time.time()
    sd = vit_config.seeded_state_dict(cfg, 0)
    vit = HipViT(cfg, sd, dev)
    print(f"weights + create: {time.time() - t0:.1f} s")
    crops = torch.randn(args.crops, 3, 224, 224, device=dev)
    ms = timeit(lambda: vit.encode(crops), iters=3, warmup=1)
    fl = 2.0 * cfg.macs_per_crop() * args.crops
    print(f"encode {args.crops} crops: {ms:8.2f} ms  {args.crops / ms * 1e3:8.1f} crops/s  {args.crops / 4 / ms * 1e3:8.1f} img/s  "
          f"{fl / ms / 1e9:8.1f} TFLOP/s  ({fl / ms / 1e9 / 2516.6 * 100:.1f}% of 2.5166 PF)")


if __name__ == "__main__":
    main()
