#!/bin/bash
# Developer: cache-side counters of the vendor GEMM next to ours (same shapes, same data), through gpurun.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/vendor_cmp2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pa -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pb -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d $OUT/pc -- python3 $R/tools/pmc_vendor_cmp.py > $OUT/pc.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/vendor_cmp2"
with open(out + "/summary.txt", "w") as fo:
    for d in sorted(glob.glob(out + "/p[abc]")):
        f = glob.glob(d + "/*/*counter_collection.csv")
        if not f: print(d, "no csv", open(d + ".log").read()[-800:], file=fo); continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"]
            if "gemm" not in k and "Cijk" not in k: continue
            key = (k[:60], r["Grid_Size"])
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in n[key]:
                n[key].add(r["Dispatch_Id"]); dur[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for key, c in acc.items():
            print(os.path.basename(d), key, "launches", len(n[key]), "avg_us %.1f" % (dur[key] / len(n[key]) / 1e3), {k: "%.4g" % (v / len(n[key])) for k, v in c.items()}, file=fo)
print(open(out + "/summary.txt").read()[-6000:])
PY
