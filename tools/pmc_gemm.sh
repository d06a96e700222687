#!/bin/bash
# Developer: LDS / issue counters of the GEMM main loop for two library builds (A/B), run through gpurun.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in cur v1; do
  if [ $v = v1 ]; then export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_v1.so; else unset CLIPENC_LIB_PATH; fi
  [ $v = v1 ] && [ ! -f $CLIPENC_LIB_PATH ] && continue
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/${v}_a -- python3 $R/tools/pmc_gemm.py > $OUT/${v}_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/${v}_b -- python3 $R/tools/pmc_gemm.py > $OUT/${v}_b.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_gemm"
for d in sorted(glob.glob(out + "/*_[ab]")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        key = (k[28:60], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[key]:
            n[key].add(r["Dispatch_Id"]); dur[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for key, c in acc.items():
        print(os.path.basename(d), key, "launches", len(n[key]), "avg_us %.1f" % (dur[key] / len(n[key]) / 1e3), {k: "%.4g" % (v / len(n[key])) for k, v in c.items()})
PY
