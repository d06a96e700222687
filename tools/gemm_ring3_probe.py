#!/usr/bin/env python3
"""Developer probe (round 3): the one-tile-per-workgroup bf16 GEMM with a two-slot ring (the shipped pipeline) against a
THREE-slot A ring (tools/probes/gemm_ring3_probe.hip), interleaved on one box, random operands: time, TFLOP/s and main-loop
shader cycles per K=32 (in-kernel stamps, second pass)."""
import ctypes, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "probes", "libgemm_ring3_probe.so"))
f = lib.gemm_ring_probe
f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
M = int(os.environ.get("GEMM_M", "262144"))
for (N, K) in ((3072, 1024), (4096, 1024), (1024, 1024), (1024, 4096)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    outs = {}
    res = {2: [], 3: []}
    for ring in (2, 3):
        o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        rc = f(ring, a.data_ptr(), w.data_ptr(), o.data_ptr(), M, N, K, None, st); torch.cuda.synchronize()
        assert rc == 0, rc
        outs[ring] = o
    idx = torch.tensor([0, 1, 255, 256, M // 2 + 3, M - 1], device=dev)
    ref = (a[idx].float() @ w.float().t())
    for ring in (2, 3):
        err = (outs[ring][idx].float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, (ring, err)
    assert torch.equal(outs[2], outs[3]), "ring 2 and ring 3 differ"        # same k order per accumulator: same bits
    o = outs[2]
    for rep in range(4):                                  # interleaved rounds
        for ring in (2, 3):
            for _ in range(2): f(ring, a.data_ptr(), w.data_ptr(), o.data_ptr(), M, N, K, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): f(ring, a.data_ptr(), w.data_ptr(), o.data_ptr(), M, N, K, None, st)
            e1.record(); torch.cuda.synchronize()
            res[ring].append(e0.elapsed_time(e1) / 8)
    tiles = ((M + 255) // 256) * (N // 256)
    line = f"N={N} K={K} M={M}:"
    for ring in (2, 3):
        stamps = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
        for _ in range(3): f(ring, a.data_ptr(), w.data_ptr(), o.data_ptr(), M, N, K, stamps.data_ptr(), st)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().astype(np.float64)
        ms = float(np.median(res[ring]))
        line += f"  ring{ring} {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF/s, loop {np.median(s[:, 1]) / (K / 32):.0f} cyc/K32 at {np.median(s[:, 1] / np.maximum(s[:, 0], 1)) / 10:.2f} GHz;"
    print(line + f"  ring3/ring2 time = {np.median(res[3]) / np.median(res[2]):.3f}", flush=True)
