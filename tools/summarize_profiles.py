#!/usr/bin/env python3
"""Developer: turn gpurun_out/<tag>/ rocprofv3 output into the committed summaries under profiles/<tag>/."""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench_n1.json"), os.path.join(dst, "bench_n1.json"))
def newest(pattern):
    """gpurun merges every run's output into the same folder: take the latest file of a kind"""
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(os.path.join(src, "stats", "*", "*kernel_stats.csv")), os.path.join(dst, "bench_n1_kernel_stats.csv"))


def shape_of(kernel_name, dispatch_ids):
    """The residual GEMM runs two shapes under one kernel name, alternating per block: out-projection (K = width), then FC2
    (K = mlp_dim).  Dispatches of that kernel in launch order: even = out-proj, odd = FC2 (the tower's last block keeps the order)."""
    order = {d: i for i, d in enumerate(sorted(dispatch_ids, key=int))}
    return lambda d: ("out_proj", "fc2")[order[d] & 1]


def per_kernel(path, names):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    rows = list(csv.DictReader(open(path)))
    resid = [k for k in {r["Kernel_Name"] for r in rows} if "gemm_persist_kernel<3, -1>" in k or "gemm_fp8_kernel<3, -1, false>" in k]
    splitters = {k: shape_of(k, {r["Dispatch_Id"] for r in rows if r["Kernel_Name"] == k}) for k in resid}
    rows_extra = []
    for r in rows:                                               # the same rows again under "shape:<which>(<kernel>)"
        if r["Kernel_Name"] in splitters:
            short = re.search(r"(\w+_kernel<[^>]*>)", r["Kernel_Name"]).group(1)
            rows_extra.append(dict(r, Kernel_Name=f"shape:{splitters[r['Kernel_Name']](r['Dispatch_Id'])}({short})"))
    for r in rows + rows_extra:
        k = r["Kernel_Name"]
        if not any(s in k for s in ("gemm", "attn", "fcreg", "head_kernel", "patchify", "embed_ln", "quant_", "row_norm")):
            continue
        if r["Counter_Name"] in names:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, {k: len(v) for k, v in disp.items()}, dur


f, nf, _ = per_kernel(newest(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv")), {"FETCH_SIZE"})
w, nw, _ = per_kernel(newest(os.path.join(src, "pmc_write", "*", "*counter_collection.csv")), {"WRITE_SIZE"})
traffic = {}
for k in f:
    # FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM): x2
    rd = 2 * f[k]["FETCH_SIZE"] * 1024 / nf[k]
    wr = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / max(nw.get(k, 1), 1)
    traffic[k] = {"launches": nf[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                  "note": "L2-miss traffic (Infinity-Cache hits are counted by the fabric counters); FETCH_SIZE x2 gfx950 correction"}
# sha of the sources each profiled kernel was built from: bench.py quotes `roofline.traffic` only while they are unchanged
sys.path.insert(0, root)
import bench  # noqa: E402
import re  # noqa: E402
short = {}
for k in traffic:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", k)
    if m:
        short[m.group(1)] = bench.kernel_source_sha(m.group(1))
# the bench run this profile sits next to: which box, what its matrix pipes get alone, and the PROBLEM the launches were profiled on
problem = None
try:
    bl = json.loads([l for l in open(os.path.join(src, "bench_n1.json")) if l.startswith("{")][-1])
    problem = bl.get("config", {}).get("problem")
    box = {"value": bl.get("value"), "ms_per_step": bl.get("ms_per_step"), "env": bl.get("env"),
           "power_capped_mfma_stream": bl.get("roofline", {}).get("power_capped_mfma_stream")}
    json.dump(box, open(os.path.join(dst, "box.json"), "w"), indent=1)
    print("box:", box["value"], "images/s;", (box["power_capped_mfma_stream"] or {}).get("value"), "TFLOP/s pure MFMA stream")
except Exception as exc:
    print("no box record:", exc)
traffic["_meta"] = {"source_sha": short, "note": "sha256[:16] of the kernel's .hip sources + common.h + gemm.h at profile time",
                    # bench.py quotes a constant only for launches of the same problem (token rows, GEMM widths)
                    "problem": problem or bench.LEGACY_PROBLEM}
json.dump(traffic, open(os.path.join(dst, "pmc_hbm_traffic_per_kernel.json"), "w"), indent=1)
del traffic["_meta"]
sq_names = {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
            "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"}
s, ns, dur = per_kernel(newest(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv")), sq_names)
with open(os.path.join(dst, "pmc_sq_summary.txt"), "w") as out:
    for k in s:
        c = s[k]
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                      # per-XCD sum / 8 = shader cycles of the launches
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024) if cycles else 0
        clk = cycles / dur[k] if dur[k] else 0                   # cycles per ns = GHz
        line = (f"{k[:100]}\n   launches {ns[k]}  MFMA-busy/(cycles x 1024 SIMDs) = {util:.3f}  clock ~{clk:.2f} GHz  "
                f"LDS bank-conflict cycles / wave cycles = {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_WAVE_CYCLES'], 1):.4f}  "
                f"wait_any {c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f}  wait_inst {c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f}  "
                f"active {c['SQ_ACTIVE_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f}\n")
        out.write(line)
        print(line, end="")
lnf = [k for k in traffic if "gemm_persist_kernel" in k and "ILi2E" in k or "<2," in k]
print("dominant-kernel traffic keys:", lnf)
