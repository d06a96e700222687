#!/usr/bin/env python3
"""Developer calibration: what the vendor library (hipBLASLt behind torch.matmul) reaches on the same bf16 shapes and
the same device -- a practical ceiling for the hand-written GEMM under this board's power management."""
import torch
dev = torch.device("cuda", 0)
M = 526336
for (N, K) in ((1024, 1024), (1024, 4096), (4096, 1024), (3072, 1024)):
    for kind in ("random", "zeros"):
        a = (torch.randn(M, K, device=dev) if kind == "random" else torch.zeros(M, K, device=dev)).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) if kind == "random" else torch.zeros(N, K, device=dev)).to(torch.bfloat16)
        for _ in range(3): o = a @ w.t()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): o = a @ w.t()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"torch.matmul bf16 M={M} N={N} K={K} {kind:6s}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF/s")
        del a, w, o
