#!/usr/bin/env python3
"""Developer diagnostic: where a persistent-GEMM tile's time goes (in-kernel 100 MHz stamps per tile, thread 0 of
each workgroup: [1] main loop start, [2] main loop end, [3] epilogue stores issued; [0] = workgroup)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# needs the diagnostic build (make -C clip_assisted_data_labeling_amd/csrc diag): the product library has no stamp hooks
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("GEMM_M", "526336"))
for (N, K, kind) in ((1024, 1024, "random"), (1024, 4096, "random"), (4096, 1024, "random"), (1024, 4096, "zeros")):
    if kind == "random":
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    else:
        a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16); w = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tiles = ((M + 255) // 256) * (N // 256)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
    for _ in range(40):                              # long enough for the clock to settle under this load
        lib.clipenc_op_gemm_nt_stamps(a.data_ptr(), w.data_ptr(), M, N, K, o.data_ptr(), stamps.data_ptr(), st)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    wg = s[:, 0]
    t = s[:, 1:4].astype(np.float64) * 0.01           # us
    main, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    per_wg = {}
    for i in range(tiles):
        per_wg.setdefault(int(wg[i]), []).append(tuple(t[i]))
    first_main, later_main, gaps = [], [], []
    for k, v in per_wg.items():
        v.sort()
        first_main.append(v[0][1] - v[0][0])
        later_main += [x[1] - x[0] for x in v[1:]]
        gaps += [v[i + 1][0] - v[i][2] for i in range(len(v) - 1)]
    span = t[:, 2].max() - t[:, 0].min()
    ideal = 2.0 * 256 * 256 * K / (2516.6e12 / 256) * 1e6
    cyc = (s[:, 5] - s[:, 4]).astype(np.float64)       # shader cycles of the main loop (s_memtime)
    ghz = np.median(cyc / np.maximum(main, 1e-9)) / 1e3
    print(f"   [{kind}] main loop: {np.median(cyc) / (K / 32):.0f} shader cycles per K=32 stage (1024 = MFMA-bound) at {ghz:.2f} GHz")
    print(f"N={N} K={K}: {tiles} tiles on {len(per_wg)} WGs, span {span:.1f} us = {2.0*M*N*K/span/1e6:.0f} TF/s; per tile (median us): "
          f"main loop {np.median(main):.2f} (first tile of a WG {np.median(first_main):.2f}, later {np.median(later_main):.2f}; "
          f"MFMA-ideal {ideal:.2f})  epilogue until stores issued {np.median(epi):.2f}  stores->next main loop {np.median(gaps):.2f}")
    # how synchronised are the workgroups: spread of the epilogue start times of the k-th tile of every WG
    kth = [sorted(v)[min(3, len(v) - 1)][1] for v in per_wg.values()]
    print(f"   4th-tile epilogue start: spread p10..p90 {np.percentile(kth, 90) - np.percentile(kth, 10):.2f} us")
    # how much of the launch do the workgroups spend waiting for the slowest one?  (what a dynamic hand-out of the last rounds could win)
    fin = np.array([max(x[2] for x in v) for v in per_wg.values()]); t_begin = t[:, 0].min()
    n_tiles = np.array([len(v) for v in per_wg.values()])
    idle = (fin.max() - fin).mean() / (fin.max() - t_begin)
    print(f"   finish of the workgroups after the first start: p10 {np.percentile(fin, 10) - t_begin:.1f}  p50 {np.median(fin) - t_begin:.1f}  p90 {np.percentile(fin, 90) - t_begin:.1f}  "
          f"max {fin.max() - t_begin:.1f} us; tiles per WG {n_tiles.min()}..{n_tiles.max()}; mean idle at the end {100 * idle:.2f} % of the launch "
          f"(of which the static tail round accounts for {100 * (n_tiles.max() - n_tiles.mean()) / n_tiles.max():.2f} %)")
