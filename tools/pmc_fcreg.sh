#!/bin/bash
# Developer: SQ counters of the store-scale regressor kernel (run through gpurun from the repo root)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fcreg; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -- python3 $R/tools/bench_fcreg.py --rows 262144 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$OUT/*/*counter_collection.csv"), key=lambda p: __import__("os").path.getmtime(p))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "fcreg" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, c in acc.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print(k[:80], "launches", len(seen[k]), "avg ms", dur[k] / len(seen[k]) / 1e6)
    print("   clock GHz %.2f  MFMA-busy/(cycles*1024) %.3f  wait_any %.2f wait_inst %.2f active %.2f  lds_conflict/wave_cycles %.4f" % (
        cyc / dur[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_LDS_BANK_CONFLICT"] / c["SQ_WAVE_CYCLES"]))
PY
