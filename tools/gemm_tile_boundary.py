#!/usr/bin/env python3
"""Developer diagnostic (diagnostic library): where the time at a TILE BOUNDARY of the LayerNorm-folded persistent GEMM goes.
In-kernel 100 MHz stamps of thread 0 per tile: [1] main loop start, [2] main loop end (wave rows re-aligned), [6] row statistics
converted + workgroup barrier, [7] first 16-row block stored (column sums / biases have arrived), [3] all stores issued."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M, K = int(os.environ.get("GEMM_M", "526336")), 1024
K0 = K
Mp = (M + 255) // 256 * 256
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
stats = torch.zeros(4, Mp, 2, device=dev)
for part in range(4):
    blk = a[:, part * 256:(part + 1) * 256].float()
    stats[part, :M, 0] = blk.sum(-1); stats[part, :M, 1] = (blk * blk).sum(-1)
SHAPES = [(3072, 1024, -1, "qkv"), (4096, 1024, 0, "fc1"), (1024, 1024, None, "out_proj"), (1024, 4096, None, "fc2")]
for (N, K, act, name) in SHAPES:
    if act is None and K != a.shape[1]:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    cs = w.float().sum(-1).contiguous(); bias = torch.randn(N, device=dev) * 0.02
    o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    tiles = (Mp // 256) * (N // 256)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
    so = torch.zeros(N // 256, Mp, 2, device=dev)
    for _ in range(30 if K == 1024 else 10):
        if act is None:      # residual GEMM: x += A.W^T + b in place (small weights keep the stream finite over the repetitions)
            _lib.check(lib.clipenc_op_gemm_resid(a.data_ptr(), w.data_ptr(), M, N, K, bias.data_ptr(), o.data_ptr(), so.data_ptr(), Mp,
                                                 stamps.data_ptr(), st), "resid")
        else:
            _lib.check(lib.clipenc_op_gemm_lnfold(a.data_ptr(), w.data_ptr(), M, N, K, cs.data_ptr(), bias.data_ptr(), stats.data_ptr(), 4, Mp,
                                                  act, o.data_ptr(), stamps.data_ptr(), st), "lnfold")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    t = s * 0.01                                       # us
    per_wg = {}
    for i in range(tiles):
        per_wg.setdefault(int(s[i, 0]), []).append(t[i])
    seg = {"main loop": [], "realign+stats+barrier [2->6]": [], "first block out [6->7]": [], "blocks 1..7 [7->3]": [], "to next main loop [3->1']": [],
           "tile period": []}
    for v in per_wg.values():
        v.sort(key=lambda r: r[1])
        for j, r in enumerate(v[1:-1], 1):
            seg["main loop"].append(r[2] - r[1]); seg["realign+stats+barrier [2->6]"].append(r[6] - r[2])
            seg["first block out [6->7]"].append(r[7] - r[6]); seg["blocks 1..7 [7->3]"].append(r[3] - r[7])
            seg["to next main loop [3->1']"].append(v[j + 1][1] - r[3]); seg["tile period"].append(v[j + 1][1] - r[1])
    cyc = np.median(s[:, 5] - s[:, 4]); ghz = cyc / (np.median(seg["main loop"]) * 1e3)
    print(f"{name}: N={N} K={K}, {tiles} tiles; main loop {cyc / (K / 32):.0f} shader cycles per K=32 at {ghz:.2f} GHz; median us per tile:")
    for k, v in seg.items():
        print(f"    {k:32s} {np.median(v):6.2f}   (p10 {np.percentile(v, 10):.2f}, p90 {np.percentile(v, 90):.2f})")
