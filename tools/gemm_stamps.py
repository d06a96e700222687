#!/usr/bin/env python3
"""Developer diagnostic: where does a GEMM tile's time go (in-kernel 100 MHz stamps per workgroup)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M, N = 65536, 4096
for K in (512, 1024, 4096):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tiles = (M // 256) * (N // 256)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        lib.clipenc_op_gemm_nt_stamps(a.data_ptr(), w.data_ptr(), M, N, K, o.data_ptr(), stamps.data_ptr(), st)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    t = s[:, :5].astype(np.float64) * 0.01           # us
    hw = s[:, 5]
    pro, main, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    print(f"K={K}: kernel span {t[:,4].max()-t[:,0].min():.1f} us; per-WG median: prologue {np.median(pro):.2f}  mainloop {np.median(main):.2f}  "
          f"epilogue(issue) {np.median(epi):.2f}  store-drain {np.median(drain):.2f}  total {np.median(t[:,4]-t[:,0]):.2f} us")
    cu = {}
    for i in range(tiles):
        cu.setdefault((int(hw[i]) & 0xFFFFFFF0, int(hw[i]) >> 32), []).append((t[i, 0], t[i, 4]))
    gaps = []
    for k, v in cu.items():
        v.sort()
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    gaps = np.array(gaps)
    print(f"   {len(cu)} distinct hw ids; WG->WG gap on a CU: median {np.median(gaps):.2f} us, p10 {np.percentile(gaps,10):.2f}, p90 {np.percentile(gaps,90):.2f}; "
          f"entry spread of first 256: {np.sort(t[:,0])[255]-t[:,0].min():.2f} us")
