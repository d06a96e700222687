#!/bin/bash
# Developer: kernel names + instruction mix of the vendor GEMM next to ours (same shapes, same data), through gpurun.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/vendor_cmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pa -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pb -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pb.log 2>&1
# TCP-side request counters (the TCC_* counters abort rocprofv3 on this image: do not add them)
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pc -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pc.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/vendor_cmp"
with open(out + "/summary.txt", "w") as fo:
    for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            print("STATS", r, file=fo)
    for f in glob.glob(out + "/trace/*/*kernel_trace.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"))
            if k not in seen:
                seen.add(k); print("KERNEL", k, file=fo)
    for d in sorted(glob.glob(out + "/p[abc]")):
        f = glob.glob(d + "/*/*counter_collection.csv")
        if not f: print(d, "no csv", file=fo); continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"]
            key = (k[:90], r["Grid_Size"], r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in n[key]:
                n[key].add(r["Dispatch_Id"]); dur[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for key, c in acc.items():
            if len(n[key]) < 4: continue
            print(os.path.basename(d), key, "launches", len(n[key]), "avg_us %.1f" % (dur[key] / len(n[key]) / 1e3), {k: "%.4g" % (v / len(n[key])) for k, v in c.items()}, file=fo)
print("\n".join(l for l in open(out + "/summary.txt").read().splitlines() if ("gemm_persist" in l or "Cijk" in l) and l.startswith("p")))
PY
