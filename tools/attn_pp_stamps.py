#!/usr/bin/env python3
"""Developer: where a wave of the ping-pong attention kernel spends its cycles (M segments, V segments, waiting at the barriers), from a
-DPP_STAMPS build (results invalid: the stamps are written over the output):  bash tools/attn_variants.sh pps="-DATTN_LONG_PP=1 -DPP_STAMPS"
   ATTN_TOK=577 ATTN_CROPS=480 python tools/attn_pp_stamps.py pps"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "clip_assisted_data_labeling_amd")
crops = int(os.environ.get("ATTN_CROPS", "480")); n_tok = int(os.environ.get("ATTN_TOK", "577"))
dev = torch.device("cuda", 0); T = crops * n_tok
g = torch.Generator(device=dev); g.manual_seed(1)
qkv = (torch.randn(T, 3072, device=dev, generator=g) * 1.5).to(torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for sfx in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(PKG, f"libclipenc_hip_{sfx}.so"))
    f = lib.clipenc_op_attention
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; f.restype = ctypes.c_int
    o = torch.zeros(T, 1024, device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        assert f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st) == 0
    torch.cuda.synchronize()
    n_tasks = crops * 16
    raw = o.view(torch.int64).flatten()[: n_tasks * 8 * 8].cpu().numpy().reshape(n_tasks, 8, 8)
    ok = raw[:, :, 5] == 0x50505354414d5053
    print(f"{sfx}: {ok.mean() * 100:.0f} % of the wave records carry the marker")
    for w in range(8):
        r = raw[:, w][ok[:, w]]
        U = r[:, 4].mean()
        print(f"  wave {w}: steps {U:5.1f}  M {np.median(r[:, 0]):8.0f}  V {np.median(r[:, 1]):8.0f}  barrier {np.median(r[:, 2]):8.0f}  total {np.median(r[:, 3]):8.0f} cycles"
              f"   per step: M {np.median(r[:, 0]) / max(U, 1):6.0f}  V {np.median(r[:, 1]) / max(U, 1):6.0f}")
