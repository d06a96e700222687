#!/usr/bin/env python3
"""Developer: the fp8 tower's LayerNorm-quantise pass alone (526 336 x 1024 bf16 rows -> e4m3 + row scales) for several library
builds on one box; outputs compared with the first build's.   python tools/bench_quant.py cur v1 ..."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "clip_assisted_data_labeling_amd")


def load(sfx):
    lib = ctypes.CDLL(os.path.join(PKG, "libclipenc_hip.so" if sfx == "cur" else f"libclipenc_hip_{sfx}.so"))
    f = lib.clipenc_op_quant_rows_fp8
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p]
    return f


def main():
    names = sys.argv[1:] or ["cur"]
    dev = torch.device("cuda", 0)
    n, k = 2048 * 257, int(os.environ.get("QUANT_K", "1024"))
    g = torch.Generator(device=dev); g.manual_seed(5)
    x = (torch.randn(n, k, device=dev, generator=g) * 3 + 0.5).to(torch.bfloat16)
    x[:, 7] *= 150.0                                          # an outlier channel
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    outs = {}
    for nm in names:
        f = load(nm)
        q = torch.zeros(n, k, dtype=torch.uint8, device=dev); sc = torch.zeros(n, device=dev)
        assert f(x.data_ptr(), 0, n, k, 1, 1e-5, q.data_ptr(), sc.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs[nm] = (q, sc)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f(x.data_ptr(), 0, n, k, 1, 1e-5, q.data_ptr(), sc.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        ms = sorted(ts)[2]
        print(f"{nm:6s} {ms:7.4f} ms   {(n * k * 3 + n * 4) / ms / 1e9:6.2f} TB/s (read bf16 + write e4m3 + scales)")
    q0, s0 = outs[names[0]]
    for nm in names[1:]:
        q1, s1 = outs[nm]
        d = (q1.to(torch.int16) - q0.to(torch.int16)).abs()
        print(f"{nm} vs {names[0]}: codes differing {(d != 0).float().mean().item():.2e} (max step {d.max().item()}), "
              f"scale max rel diff {((s1 - s0).abs() / s0.abs()).max().item():.2e}")


if __name__ == "__main__":
    main()
