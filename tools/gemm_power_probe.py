#!/usr/bin/env python3
"""Developer diagnostic: is the GEMM main loop schedule-bound or clock/power-bound?  Same kernel, same shape, operands
random vs all-zero (zero operands toggle almost nothing in the MFMA datapath, so the chip holds a higher clock)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
from bench import EnvSampler
import numpy as np
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = 526336
for (N, K) in ((1024, 1024), (1024, 4096), (4096, 1024)):
    for kind in ("random", "zeros"):
        if kind == "random":
            a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        else:
            a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16); w = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def run():
            lib.clipenc_op_gemm_nt(a.data_ptr(), w.data_ptr(), M, N, K, 0, 1, None, o.data_ptr(), st)
        for _ in range(5): run()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        with EnvSampler(0) as es:                                # ~1 s of back-to-back launches under the hwmon sampler
            for _ in range(max(20, int(1000 / ms))): run()
            torch.cuda.synchronize()
        pw, fq = es.samples["power_w"], es.samples["sclk_mhz"]
        w_ = float(np.median(pw[len(pw) // 3:])) if pw else float("nan")
        f_ = float(np.median(fq[len(fq) // 3:])) if fq else float("nan")
        tf = 2.0 * M * N * K / ms / 1e9
        print(f"N={N} K={K} {kind:6s}: {ms:.3f} ms  {tf:.0f} TF/s   {w_:.0f} W  {f_:.0f} MHz  {w_ / tf:.3f} pJ/FLOP all-in")
