"""Diagnostic: accumulation error of the fp8 MFMA path relative to sum |a||w| (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_fp8 import _rand8, _gemm8, _deq

gpu = torch.device("cuda", 0)
for (m, n, k) in [(256, 256, 256), (1285, 768, 1024), (1024, 1024, 4096)]:
    g = torch.Generator().manual_seed(m + 3 * n + 7 * k)
    a8, sa = _rand8(m, k, g)
    w8, sw = _rand8(n, k, g)
    bias = torch.zeros(n)
    A, W = _deq(a8).double(), _deq(w8).double()
    scale = sa.double().view(m, 1) * sw.double().view(1, n)
    ref = (A @ W.t()) * scale
    S = (A.abs() @ W.abs().t()) * scale
    out = _gemm8(gpu, a8.to(gpu), w8.to(gpu), sa.to(gpu), sw.to(gpu), bias.to(gpu)).double().cpu()
    err = ((out - ref).abs() - ref.abs() * 2.0 ** -8).clamp(min=0)
    r = err / S
    print((m, n, k), "max excess err / sum|a||w| = %.3e  (2^%.1f)   mean %.3e" % (r.max().item(), torch.log2(r.max()).item(), r.mean().item()))
    # bf16 reference path on the same dequantised operands for comparison
    o2 = ((A.float().to(gpu).to(torch.bfloat16).float() @ W.float().to(gpu).to(torch.bfloat16).float().t()).double().cpu() * scale)
    r2 = ((o2 - ref).abs()) / S
    print("    torch fp32 matmul of the same operands: max err / sum|a||w| = %.3e" % r2.max().item())
