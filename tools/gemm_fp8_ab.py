#!/usr/bin/env python3
"""Developer: the four fp8 block GEMMs of ViT-L/14 at the bench's M, one library build per process
(CLIPENC_LIB_PATH selects it), random e4m3 operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("AB_M", "526336"))
def rnd8(r, c):
    return (torch.randn(r, c, device=dev) * 0.5).to(torch.float8_e4m3fn)
SHAPES = (("qkv", 3072, 1024, "bf16"), ("out_proj", 1024, 1024, "resid"), ("fc1", 4096, 1024, "q"), ("fc2", 1024, 4096, "resid"),
          # the fused tower's forms of the same four (block-exponent rows, include/clipenc.h)
          ("qkv_lnf", 3072, 1024, "lnf"), ("out_proj_q", 1024, 1024, "resid_q"), ("fc1_lnf", 4096, 1024, "lnf_q"),
          ("fc2_q", 1024, 4096, "resid_q"))
ONLY = os.environ.get("AB_ONLY")
for name, N, K, kind in SHAPES:
    if ONLY and name not in ONLY.split(","):
        continue
    a = rnd8(M, K); w = rnd8(N, K)
    sa = torch.rand(M, device=dev) + 0.5; sw = torch.exp2(torch.randint(-9, -6, (N,), device=dev).float()); bias = torch.randn(N, device=dev)
    inv = torch.rand(N, device=dev) + 0.5
    if kind == "q":
        out = torch.empty(M, N, device=dev, dtype=torch.uint8)
        run = lambda: lib.clipenc_op_gemm_fp8_q(a.data_ptr(), w.data_ptr(), M, N, K, None, sw.data_ptr(), bias.data_ptr(), 0, inv.data_ptr(), out.data_ptr(), st)
    elif kind in ("lnf", "lnf_q"):
        a = (torch.randn(M, K, device=dev) * 60).clamp(-255, 255).to(torch.float8_e4m3fn)
        eb = torch.randint(120, 130, (M, 4), device=dev, dtype=torch.uint8)
        rr = torch.rand(M, device=dev) + 0.5; rd = torch.randn(M, device=dev); cs = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.uint8 if kind == "lnf_q" else torch.bfloat16)
        run = lambda: lib.clipenc_op_gemm_fp8_lnf(a.data_ptr(), eb.data_ptr(), w.data_ptr(), M, N, K, rr.data_ptr(), rd.data_ptr(), sw.data_ptr(),
                                                  cs.data_ptr(), bias.data_ptr(), 0 if kind == "lnf_q" else -1,
                                                  inv.data_ptr() if kind == "lnf_q" else None, out.data_ptr(), st)
    elif kind == "resid_q":
        out = torch.randn(M, N, device=dev).to(torch.bfloat16)
        q8 = torch.empty(M, N, device=dev, dtype=torch.uint8); eb = torch.empty(M, 4, device=dev, dtype=torch.uint8)
        ld = (M + 255) // 256 * 256
        stt = torch.empty(N // 256, ld, 2, device=dev)
        run = lambda: lib.clipenc_op_gemm_fp8_resid_q(a.data_ptr(), w.data_ptr(), M, N, K, sw.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                      q8.data_ptr(), eb.data_ptr(), stt.data_ptr(), ld, st)
    elif kind == "resid":
        out = torch.randn(M, N, device=dev).to(torch.bfloat16)
        run = lambda: lib.clipenc_op_gemm_fp8(a.data_ptr(), w.data_ptr(), M, N, K, None, sw.data_ptr(), bias.data_ptr(), -1, out.data_ptr(), out.data_ptr(), st)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        run = lambda: lib.clipenc_op_gemm_fp8(a.data_ptr(), w.data_ptr(), M, N, K, sa.data_ptr(), sw.data_ptr(), bias.data_ptr(), -1, None, out.data_ptr(), st)
    for _ in range(5): assert run() == 0
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{os.environ.get('CLIPENC_LIB_PATH', 'cur')[-12:]:>12s} {name:8s} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF/s", flush=True)
    del a, w, out
