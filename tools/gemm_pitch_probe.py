#!/usr/bin/env python3
"""Developer (diagnostic library): does the ROW PITCH of the GEMM operands matter?  With K-contiguous rows of 1024 / 4096 bf16 the rows of
a tile are 2 KiB / 8 KiB apart: every 128-byte line of an 8-row LDS-DMA piece, and the same k position of every row of a panel, has the
same low address bits.  If the L2's channel selection (or the Infinity Cache's, or the HBM's) folds on those bits, the pieces of a stage
queue up on a few channels.  Same GEMM (M = 526 336, random bf16 operands), operands stored with pitch K + pad elements, interleaved."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("GEMM_M", "526336"))
pads = [int(x) for x in os.environ.get("PADS", "0,64,128,192,32").split(",")]
for (N, K) in ((3072, 1024), (1024, 4096), (1024, 1024)):
    bufs = {}
    for pad in pads:
        a = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K + pad, device=dev).to(torch.bfloat16)
        o = torch.empty(M, N + (pad if os.environ.get("PAD_OUT") else 0), device=dev, dtype=torch.bfloat16)
        bufs[pad] = (a, w, o)
    ref = None
    for rnd in range(3):
        for pad in pads:
            a, w, o = bufs[pad]
            def run():
                _lib.check(lib.clipenc_op_gemm_nt_ld(a.data_ptr(), K + pad, w.data_ptr(), K + pad, M, N, K, o.data_ptr(), o.shape[1], st), "gemm")
            for _ in range(3): run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"N={N} K={K} round {rnd} pitch K+{pad:3d}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
    del bufs
    torch.cuda.empty_cache()
