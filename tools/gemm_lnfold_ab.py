#!/usr/bin/env python3
"""Developer: the LayerNorm-folded GEMM (QKV / FC1 shapes) on fixed random operands, for same-box A/B of diagnostic
library builds:   CLIPENC_LIB_PATH=.../libclipenc_hip_<x>.so python tools/gemm_lnfold_ab.py
(needs a -DCLIPENC_DIAG build: clipenc_op_gemm_lnfold is not part of the product ABI)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M, K = 526336, 1024
Mp = (M + 255) // 256 * 256
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
af = a.float()
stats = torch.zeros(4, Mp, 2, device=dev)
for part in range(4):
    blk = af[:, part * 256:(part + 1) * 256]
    stats[part, :M, 0] = blk.sum(-1); stats[part, :M, 1] = (blk * blk).sum(-1)
del af
res = {}
for (N, act, name) in ((3072, -1, "qkv"), (4096, 0, "fc1")):
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    cs = w.float().sum(-1).contiguous(); bias = torch.randn(N, device=dev) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run():
        _lib.check(lib.clipenc_op_gemm_lnfold(a.data_ptr(), w.data_ptr(), M, N, K, cs.data_ptr(), bias.data_ptr(), stats.data_ptr(), 4, Mp,
                                              act, o.data_ptr(), None, st), "lnfold")
    for _ in range(5): run()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    # sampled rows against a float reference (same folded arithmetic)
    idx = torch.tensor([0, 1, 255, 256, 70000, M - 257, M - 1], device=dev)
    x = a[idx].float(); mean = x.mean(-1, keepdim=True); var = (x * x).mean(-1, keepdim=True) - mean * mean
    ref = (x - mean) * torch.rsqrt(var + 1e-5) @ w.float().t() + bias
    if act == 0: ref = ref * torch.sigmoid(1.702 * ref)
    err = (o[idx].float() - ref).abs().max().item()
    print(f"{name}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s  max err on sampled rows {err:.3f}")
