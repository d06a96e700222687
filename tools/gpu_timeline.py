#!/usr/bin/env python3
"""Developer: GPU busy / idle time and the largest gaps from a rocprofv3 --kernel-trace csv (which kernels ran when).
    python tools/gpu_timeline.py gpurun_out/prof_e2e"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
busy, (cs, ce, _) = 0, iv[0]
gaps = []
last_name = iv[0][2]
for s, e, n in iv[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, (ce - t0) / 1e9, last_name[:50], n[:50])); cs, ce = s, e
    else:
        ce = max(ce, e)
    last_name = n
busy += ce - cs
print(f"span {(t1 - t0) / 1e9:.3f} s  busy {busy / 1e9:.3f} s  idle {(t1 - t0 - busy) / 1e9:.3f} s")
for g, t, a, b in sorted(gaps, reverse=True)[:10]:
    print(f"  gap {g / 1e6:8.2f} ms at {t:.3f} s  after {a}  before {b}")
tot = collections.defaultdict(float)
for s, e, n in iv:
    k = ("encoder" if any(x in n for x in ("gemm", "attn", "patchify", "embed_ln", "head_k", "gather_row", "fcreg")) else
         "jpeg" if "jpeg" in n else "preproc" if "preproc" in n else "copy/fill" if "rocclr" in n else "other")
    tot[k] += (e - s) / 1e9
print({k: round(v, 3) for k, v in tot.items()})
