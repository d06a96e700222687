#!/usr/bin/env python3
"""Developer: board power, sustained shader clock and joules per instruction for single instruction types on all 256 CUs
(tools/probes/power_probe.hip).  The encoder runs at the 1400 W cap, so a kernel's time follows its energy."""
import ctypes, os, sys, time
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bench import EnvSampler  # noqa: E402

ROLES = [("sleep", 0, 0, ""), ("mfma 32x32x16 bf16", 1, 8, "mfma"), ("mfma 32x32x16 zeros", 9, 8, "mfma"), ("mfma 16x16x32 bf16", 2, 16, "mfma"),
         ("mfma 16x16x32 zeros", 10, 16, "mfma"), ("v_exp_f32", 3, 16, "instr"), ("v_fma_f32", 4, 64, "instr"), ("v_max3_f32", 8, 64, "instr"),
         ("v_dot2c_f32_bf16", 11, 64, "instr"), ("ds_read_b128", 5, 16, "instr"), ("ds_read_b64_tr_b16", 6, 16, "instr"),
         ("mfma32 + 2 exp", 7, 8, "mfma"), ("mfma32 + ds_read_b128", 12, 8, "mfma"),
         ("mfma_scale 32x32x64 fp8", 13, 4, "mfma"), ("mfma_scale 32x32x64 zeros", 15, 4, "mfma"), ("mfma_scale 16x16x128 fp8", 14, 8, "mfma"),
         # issue order of a 4 x 4 block of 16x16x32 MFMAs (PROBE_ONLY=order runs just these, three interleaved rounds)
         ("4x4 block, shipped order (srcA every MFMA, srcB every 4th)", 16, 16, "order"), ("4x4 block, serpentine (one operand per MFMA)", 17, 16, "order"),
         ("4x4 block, diagonal (both operands every MFMA)", 18, 16, "order"), ("4x4 block, transposed (srcB every MFMA, srcA every 4th)", 19, 16, "order"),
         ("4x4 block, transposed serpentine", 21, 16, "order"), ("4x4 block, one operand pair for all 16", 20, 16, "order"),
         # round 4 (PROBE_ONLY=acc): the accumulator port
         ("one operand pair AND one accumulator for all 16", 22, 16, "acc"), ("4x4 block, one operand pair, 16 accumulators", 20, 16, "acc"),
         ("2 k halves, shipped: k half outer, serpentine (acc changes every MFMA)", 23, 32, "acc"),
         ("2 k halves back to back per accumulator (k 0,1 | 0,1)", 24, 32, "acc"),
         ("2 k halves back to back, alternating (k 0,1 | 1,0)", 25, 32, "acc")]


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "libpower_probe.so"))
    f = lib.power_probe_run
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    rnd = torch.cat([torch.randn(8192, device=dev, generator=g).to(torch.bfloat16).view(torch.int32),       # 4096 dwords of bf16 pairs
                     (torch.randn(16384, device=dev, generator=g) * 2).to(torch.float8_e4m3fn).view(torch.int32)]).contiguous()   # + 4096 of e4m3
    out = torch.zeros(256 * 8, device=dev, dtype=torch.int64)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    secs = float(os.environ.get("PROBE_SECONDS", "1.5"))
    base_w = None
    print(f"{'role':60s} {'memtime GHz':>11s} {'hwmon MHz':>9s} {'W':>7s} {'cyc/instr':>9s} {'G instr/s':>10s} {'nJ/instr (above sleep)':>22s}")
    only = os.environ.get("PROBE_ONLY", "")
    roles = [r for r in ROLES if not only or r[3] == only or r[1] == 0]
    if only:
        roles = roles[:1] + roles[1:] * int(os.environ.get("PROBE_ROUNDS", "3"))      # interleaved rounds: same-box, same-minute A/B
    for name, role, per_body, kind in roles:
        f(role, rnd.data_ptr(), out.data_ptr(), 2000, st); torch.cuda.synchronize()
        t0 = time.perf_counter(); f(role, rnd.data_ptr(), out.data_ptr(), 20000, st); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        iters = int(min(2e9, max(1000, 20000 * secs / dt)))
        with EnvSampler(0) as es:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(role, rnd.data_ptr(), out.data_ptr(), iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        cyc = out.double().mean().item()
        ghz = cyc / ms / 1e6
        pw = es.samples["power_w"]
        w = float(np.median(pw[len(pw) // 3:])) if pw else float("nan")
        fq = es.samples["sclk_mhz"]
        mhz = float(np.median(fq[len(fq) // 3:])) if fq else float("nan")
        if role == 0:
            base_w = w
            print(f"{name:60s} {ghz:11.3f} {mhz:9.0f} {w:7.0f}")
            continue
        n_instr = per_body * iters * 256.0 * 8
        rate = n_instr / (ms * 1e-3)
        nj = (w - base_w) / rate * 1e9
        print(f"{name:60s} {ghz:11.3f} {mhz:9.0f} {w:7.0f} {cyc / (per_body * iters):9.2f} {rate / 1e9:10.1f} {nj:22.2f}")


if __name__ == "__main__":
    main()
