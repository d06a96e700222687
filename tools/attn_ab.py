#!/usr/bin/env python3
"""Developer: the attention kernel alone (ViT-L/14 shape: 2048 crops x 257 tokens x 16 heads) for several library builds on ONE box,
interleaved; every build's output is compared with the first one's.
  python tools/attn_ab.py cur a1 a2 ...     ("cur" = libclipenc_hip.so, else libclipenc_hip_<suffix>.so, see tools/attn_variants.sh)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "clip_assisted_data_labeling_amd")


def load(sfx):
    lib = ctypes.CDLL(os.path.join(PKG, "libclipenc_hip.so" if sfx == "cur" else f"libclipenc_hip_{sfx}.so"))
    f = lib.clipenc_op_attention
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    f.restype = ctypes.c_int
    return f


def main():
    names = sys.argv[1:] or ["cur"]
    crops = int(os.environ.get("ATTN_CROPS", "2048")); n_tok = int(os.environ.get("ATTN_TOK", "257"))
    dev = torch.device("cuda", 0)
    T = crops * n_tok
    g = torch.Generator(device=dev); g.manual_seed(1)
    qkv = (torch.randn(T, 3072, device=dev, generator=g) * 1.5).to(torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    fns = {n: load(n) for n in names}
    outs = {}
    for n, f in fns.items():
        o = torch.zeros(T, 1024, device=dev, dtype=torch.bfloat16)
        assert f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st) == 0
        torch.cuda.synchronize()
        outs[n] = o
    ref = outs[names[0]].float()
    for n in names[1:]:
        d = (outs[n].float() - ref).abs().max().item()
        print(f"{n}: max |out - {names[0]}| = {d:.3e}  (ref max {ref.abs().max().item():.3f})  equal bits: {bool((outs[n] == outs[names[0]]).all())}")
    o = torch.empty(T, 1024, device=dev, dtype=torch.bfloat16)
    res = {n: [] for n in names}
    for rep in range(5):
        for n, f in fns.items():
            for _ in range(3):
                f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(24):
                f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st)
            e.record(); torch.cuda.synchronize()
            res[n].append(s.elapsed_time(e) / 24)
    fl = 4.0 * crops * 16 * n_tok * n_tok * 64
    env = {}
    if os.environ.get("ATTN_POWER"):                       # ~1.5 s of back-to-back launches per build under the hwmon sampler
        sys.path.insert(0, ROOT)
        from bench import EnvSampler
        import numpy as np
        for n, f in fns.items():
            reps = int(1500 / (sorted(res[n])[2]))
            with EnvSampler(0) as es:
                for _ in range(reps):
                    f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st)
                torch.cuda.synchronize()
            pw, fq = es.samples["power_w"], es.samples["sclk_mhz"]
            env[n] = f"  {np.median(pw[len(pw) // 3:]):6.0f} W  hwmon sclk {np.median(fq[len(fq) // 3:]):6.0f} MHz"
    for n in names:
        v = sorted(res[n]); med = v[len(v) // 2]
        print(f"{n:6s} median {med:7.4f} ms  (min {v[0]:.4f} max {v[-1]:.4f})  {fl / med / 1e9:7.1f} TFLOP/s   x24 layers = {24 * med:6.2f} ms/step{env.get(n, '')}")


if __name__ == "__main__":
    main()
