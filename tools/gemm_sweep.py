#!/usr/bin/env python3
"""Developer sweep: GEMM time vs K (slope = main-loop rate, intercept = per-tile prologue+epilogue)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
from tools.quick_bench import timeit

lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("M", 65536)); N = int(os.environ.get("N", 4096))
iters = int(os.environ.get("ITERS", 10))
for epi, name in ((1, "bf16"), (0, "f32")):
    for K in (128, 256, 512, 1024, 2048, 4096):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if epi == 1 else torch.float32)
        ms = timeit(lambda: lib.clipenc_op_gemm_nt(a.data_ptr(), w.data_ptr(), M, N, K, 0, epi, None, o.data_ptr(), st), iters=iters)
        tiles_per_cu = (M // 256) * (N // 256) / 256
        print(f"epi={name} M={M} N={N} K={K}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s  {ms*1e3/tiles_per_cu:7.2f} us/tile", flush=True)
