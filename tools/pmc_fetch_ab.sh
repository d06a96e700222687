#!/bin/bash
# Developer: L2-miss read / write bytes per launch of the encoder's GEMMs for several library builds on one box (rocprofv3 PMC,
# separate FETCH_SIZE / WRITE_SIZE passes; gfx950: FETCH_SIZE x2).   bash tools/pmc_fetch_ab.sh cur sg16 ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fetch_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = cur ]; then unset CLIPENC_LIB_PATH; else export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${v}_$c
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${v}_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/${v}_$c.log 2>&1
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, os, sys
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_fetch_ab"
for v in sys.argv[1:]:
    tot = collections.defaultdict(lambda: [0.0, 0.0, set()])
    for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        f = glob.glob(f"{out}/{v}_{c}/*/*counter_collection.csv")
        if not f: print(v, c, "no csv"); continue
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"]
            if "gemm_persist" not in k and "attn" not in k: continue
            k = k[k.find("gemm_persist") if "gemm_persist" in k else k.find("attn"):][:36]
            if r["Counter_Name"] == c:
                tot[k][ci] += float(r["Counter_Value"]) * 1024 * (2 if ci == 0 else 1)
                if ci == 0: tot[k][2].add(r["Dispatch_Id"])
    for k, (rd, wr, d) in sorted(tot.items()):
        n = max(len(d), 1)
        print(f"{v:6s} {k:38s} launches {n:4d} read {rd / n / 1e9:7.3f} GB write {wr / n / 1e9:7.3f} GB per launch")
PY
