#!/usr/bin/env python3
"""Developer diagnostic (needs `make diag`): section-by-section shader-cycle timeline of the skewed-epilogue GEMM around a
tile boundary, for wave row X (wave 0) and wave row Y (wave 4), median over workgroups.
Stamp points (gemm_persist.hip, SKEW path): tail stage L: 20 section start | 21 PA_L wait done | 22 barrier passed | 23 MFMAs
issued | 24 S1 start | 25 6 pieces issued | 26 epilogue half 0 done | 27 reads + vmcnt(16) done | 28 barrier passed | 29
MFMAs issued | 30 = next tile; first pair: 0 S2 start | 1 2 pieces | 2 epilogue half 1 done | 3 reads + vmcnt(24) | 4 barrier
| 5 MFMAs | 6 PB_0' start | 7 wait done | 8 barrier | 9 MFMAs | 10 | 11 PA_1' wait | 12 | 13 | 14 PB_1' wait | 15 | 16 | 17."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M, K = 131584, 1024
Mp = (M + 255) // 256 * 256
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
stats = torch.zeros(4, Mp, 2, device=dev)
for part in range(4):
    blk = a[:, part * 256:(part + 1) * 256].float()
    stats[part, :M, 0] = blk.sum(-1); stats[part, :M, 1] = (blk * blk).sum(-1)
for (N, act, name) in ((3072, -1, "qkv"), (4096, 0, "fc1")):
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    cs = w.float().sum(-1).contiguous(); bias = torch.randn(N, device=dev) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    stamps = torch.zeros(256 * 128, dtype=torch.int64, device=dev)
    for _ in range(30):
        _lib.check(lib.clipenc_op_gemm_lnfold(a.data_ptr(), w.data_ptr(), M, N, K, cs.data_ptr(), bias.data_ptr(), stats.data_ptr(), 4, Mp,
                                              act, o.data_ptr(), stamps.data_ptr(), st), "lnfold")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(256, 2, 2, 32).astype(np.int64)      # [wg][row][tile 2|3][stamp]
    print(f"--- {name} (N={N}): shader cycles, median over workgroups")
    for row, rn in ((0, "X"), (1, "Y")):
        t2, t3 = s[:, row, 0, :], s[:, row, 1, :]
        def d(a_, b_):
            return int(np.median((b_ - a_) & 0xffffffff))
        tail = [("PA_L sect", d(t2[:, 20], t2[:, 21])), ("->barrier", d(t2[:, 21], t2[:, 22])), ("mfma", d(t2[:, 22], t2[:, 23])),
                ("->S1", d(t2[:, 23], t2[:, 24])), ("S1 6 pieces", d(t2[:, 24], t2[:, 25])), ("S1 epilogue", d(t2[:, 25], t2[:, 26])),
                ("S1 reads+vmcnt16", d(t2[:, 26], t2[:, 27])), ("->barrier", d(t2[:, 27], t2[:, 28])), ("mfma", d(t2[:, 28], t2[:, 29])),
                ("->S2", d(t2[:, 29], t3[:, 0])),
                ("S2 2 pieces", d(t3[:, 0], t3[:, 1])), ("S2 epilogue", d(t3[:, 1], t3[:, 2])), ("S2 reads+vmcnt24", d(t3[:, 2], t3[:, 3])),
                ("->barrier", d(t3[:, 3], t3[:, 4])), ("mfma", d(t3[:, 4], t3[:, 5])), ("->PB_0'", d(t3[:, 5], t3[:, 6])),
                ("PB_0' sect (vmcnt24)", d(t3[:, 6], t3[:, 7])), ("->barrier", d(t3[:, 7], t3[:, 8])), ("mfma", d(t3[:, 8], t3[:, 9])),
                ("PA_1' sect (vmcnt16)", d(t3[:, 10], t3[:, 11])), ("->barrier", d(t3[:, 11], t3[:, 12])), ("mfma", d(t3[:, 12], t3[:, 13])),
                ("PB_1' sect (vmcnt8)", d(t3[:, 13], t3[:, 14])), ("->barrier", d(t3[:, 14], t3[:, 15])), ("mfma", d(t3[:, 15], t3[:, 16]))]
        print(f"  row {rn}: " + " | ".join(f"{k} {v}" for k, v in tail))
        print(f"  row {rn}: tile 2 S1 start -> tile 3 stage 1 end: {d(t2[:, 24], t3[:, 17])} cycles; whole tile 3 (stamp 0 -> 30): {d(t3[:, 0], t3[:, 30])}")
