#!/usr/bin/env python3
"""Developer diagnostic (diagnostic library): the tile timeline of the fp8 block GEMMs of the fused tower at the bench's M.
In-kernel stamps of thread 0 per tile: [1] main loop start, [2] main loop end (wave rows re-aligned), [3] epilogue done (all stores
issued); shader cycles over the main loop give the in-kernel clock and the cycles per K = 128 stage (2 048 = MFMA-bound)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIPENC_LIB_PATH", os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so"))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
M = int(os.environ.get("GEMM_M", "526336")); Mp = (M + 255) // 256 * 256
ZERO = os.environ.get("GEMM_ZERO") == "1"           # all-zero operands: nothing toggles in the MFMA datapath, the schedule alone shows
def rnd8(r, c, s=0.5):
    t = torch.zeros(r, c, device=dev) if ZERO else torch.randn(r, c, device=dev) * s
    return t.clamp(-255, 255).to(torch.float8_e4m3fn)
SHAPES = (("qkv_lnf", 3072, 1024, "lnf"), ("fc1_lnf", 4096, 1024, "lnf_q"), ("out_proj_q", 1024, 1024, "resid_q"), ("fc2_q", 1024, 4096, "resid_q"),
          ("fc1_plain_q", 4096, 1024, "q"), ("fc2_plain", 1024, 4096, "resid"))
ONLY = os.environ.get("AB_ONLY")
for name, N, K, kind in SHAPES:
    if ONLY and name not in ONLY.split(","):
        continue
    a = rnd8(M, K, 60 if kind.startswith("lnf") else 0.5); w = rnd8(N, K)
    sw = torch.exp2(torch.randint(-9, -6, (N,), device=dev).float()); bias = torch.randn(N, device=dev); inv = torch.rand(N, device=dev) + 0.5
    tiles = (Mp // 256) * (N // 256)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
    if kind in ("lnf", "lnf_q"):
        eb = torch.randint(120, 130, (M, 4), device=dev, dtype=torch.uint8)
        rr = torch.rand(M, device=dev) + 0.5; rd = torch.randn(M, device=dev); cs = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.uint8 if kind == "lnf_q" else torch.bfloat16)
        run = lambda: lib.clipenc_op_gemm_fp8_lnf(a.data_ptr(), eb.data_ptr(), w.data_ptr(), M, N, K, rr.data_ptr(), rd.data_ptr(), sw.data_ptr(),
                                                  cs.data_ptr(), bias.data_ptr(), 0 if kind == "lnf_q" else -1,
                                                  inv.data_ptr() if kind == "lnf_q" else None, out.data_ptr(), st)
    elif kind == "resid_q":
        out = torch.randn(M, N, device=dev).to(torch.bfloat16)
        q8 = torch.empty(M, N, device=dev, dtype=torch.uint8); eb = torch.empty(M, 4, device=dev, dtype=torch.uint8)
        stt = torch.empty(N // 256, Mp, 2, device=dev)
        run = lambda: lib.clipenc_op_gemm_fp8_resid_q(a.data_ptr(), w.data_ptr(), M, N, K, sw.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                      q8.data_ptr(), eb.data_ptr(), stt.data_ptr(), Mp, st)
    elif kind == "q":
        out = torch.empty(M, N, device=dev, dtype=torch.uint8)
        run = lambda: lib.clipenc_op_gemm_fp8_q(a.data_ptr(), w.data_ptr(), M, N, K, None, sw.data_ptr(), bias.data_ptr(), 0, inv.data_ptr(), out.data_ptr(), st)
    else:
        out = torch.randn(M, N, device=dev).to(torch.bfloat16)
        run = lambda: lib.clipenc_op_gemm_fp8(a.data_ptr(), w.data_ptr(), M, N, K, None, sw.data_ptr(), bias.data_ptr(), -1, out.data_ptr(), out.data_ptr(), st)
    _lib.check(lib.clipenc_diag_fp8_stamps(None), "stamps off")
    for _ in range(20): _lib.check(run(), name)                     # warm: the board settles at its sustained clock
    s0 = torch.cuda.Event(enable_timing=True); e0 = torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10): run()
    e0.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(e0) / 10
    _lib.check(lib.clipenc_diag_fp8_stamps(stamps.data_ptr()), "stamps on")
    for _ in range(3): _lib.check(run(), name)
    torch.cuda.synchronize()
    _lib.check(lib.clipenc_diag_fp8_stamps(None), "stamps off")
    s = stamps.cpu().numpy().astype(np.float64); t = s * 0.01      # us
    per_wg = {}
    for i in range(tiles):
        per_wg.setdefault(int(s[i, 0]), []).append(t[i])
    seg = {"main loop": [], "epilogue [2->3]": [], "to next main loop [3->1']": [], "tile period": []}
    for v in per_wg.values():
        v.sort(key=lambda r: r[1])
        for j, r in enumerate(v[1:-1], 1):
            seg["main loop"].append(r[2] - r[1]); seg["epilogue [2->3]"].append(r[3] - r[2])
            seg["to next main loop [3->1']"].append(v[j + 1][1] - r[3]); seg["tile period"].append(v[j + 1][1] - r[1])
    cyc = np.median(s[:, 5] - s[:, 4]); ghz = cyc / (np.median(seg["main loop"]) * 1e3)
    print(f"{name}: N={N} K={K}{' zeros' if ZERO else ''}, {tiles} tiles, {ms:.3f} ms per launch = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s; main loop "
          f"{cyc / (K / 128):.0f} shader cycles per K=128 stage at {ghz:.2f} GHz; median us per tile:", flush=True)
    for k, v in seg.items():
        print(f"    {k:28s} {np.median(v):6.2f}   (p10 {np.percentile(v, 10):.2f}, p90 {np.percentile(v, 90):.2f})", flush=True)
    del a, w, out
