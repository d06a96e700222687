#!/usr/bin/env python3
"""Developer: section timeline of the attention stream kernel built with -DATTN_STAMPS (tools/attn_variants.sh s0="-DATTN_STAMPS" ...):
per compute wave, shader cycles per 32-query block spent in  wait-for-K/V | QK MFMAs | max | P.V | epilogue, and blocks per wave.
  python tools/attn_stamps.py s0 s1 ..."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "clip_assisted_data_labeling_amd")
NCW = 7


def main():
    crops, n_tok = 2048, 257
    dev = torch.device("cuda", 0)
    T = crops * n_tok
    g = torch.Generator(device=dev); g.manual_seed(1)
    qkv = (torch.randn(T, 3072, device=dev, generator=g) * 1.5).to(torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.join(PKG, f"libclipenc_hip_{n}.so"))
        f = lib.clipenc_op_attention
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        o = torch.zeros(T, 1024, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            assert f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st) == 0
        torch.cuda.synchronize()
        raw = o.view(torch.int64).flatten()[: 256 * NCW * 8].cpu().view(256, NCW, 8)
        assert (raw[:, :, 7] == 0x5741505354414d50).all(), "stamp marker missing"
        sect = raw[:, :, :5].double(); nblk = raw[:, :, 5].double(); tot = raw[:, :, 6].double()
        print(f"== {n}: kernel span per wave (mean) {tot.mean().item():.0f} cycles; blocks per workgroup {nblk.sum(1).mean().item():.0f}")
        print("   wave  blocks   cycles/block:  wait     QK    max     PV    epi    sum   | wave total")
        for w in range(NCW):
            nb = nblk[:, w].mean().item()
            per = (sect[:, w, :].sum(0) / nblk[:, w].sum()).tolist()
            print(f"   {w:4d} {nb:7.1f}               " + " ".join(f"{v:6.0f}" for v in per) + f" {sum(per):6.0f}   | {tot[:, w].mean().item():9.0f}")
        per = (sect.sum((0, 1)) / nblk.sum()).tolist()
        print("    all                       " + " ".join(f"{v:6.0f}" for v in per) + f" {sum(per):6.0f}")
        # how far apart do the workgroups finish?  (static task ranges: what a ticketed hand-out of the tasks could win)
        wg = tot.max(1).values
        print(f"    workgroup spans (cycles): min {wg.min().item():.0f}  p50 {wg.median().item():.0f}  max {wg.max().item():.0f};  "
              f"mean idle behind the slowest workgroup {100 * (1 - wg.mean().item() / wg.max().item()):.2f} %")


if __name__ == "__main__":
    main()
