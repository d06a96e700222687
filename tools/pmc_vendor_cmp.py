#!/usr/bin/env python3
"""Developer: this library's GEMM and the vendor library's (hipBLASLt behind torch.matmul) on the same bf16 shapes and
data, a few launches each, for rocprofv3 kernel-trace / --pmc passes (tools/pmc_vendor_cmp.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("CMP_M", "131072"))
for (N, K) in ((4096, 1024), (1024, 4096), (1024, 1024)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(8):
        lib.clipenc_op_gemm_nt(a.data_ptr(), w.data_ptr(), M, N, K, 0, 1, None, o.data_ptr(), st)
    torch.cuda.synchronize()
    for _ in range(8):
        o2 = a @ w.t()
    torch.cuda.synchronize()
    del a, w, o, o2
