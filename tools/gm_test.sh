R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/gm_bench.json 2>/dev/null
rm -rf $R/gpurun_out/gm_fetch
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/gm_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
d=json.load(open(R+"/gpurun_out/gm_bench.json")); print("images/s", d["value"], {k:v for k,v in d["kernels_tflops"].items() if "persist" in k})
f=max(glob.glob(R+"/gpurun_out/gm_fetch/*/*counter_collection.csv"), key=os.path.getmtime)
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if "gemm_persist" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
        acc[r["Kernel_Name"][28:58]]+=float(r["Counter_Value"]); n[r["Kernel_Name"][28:58]].add(r["Dispatch_Id"])
for k in acc: print(k, "read GB/launch %.2f" % (2*acc[k]*1024/len(n[k])/1e9))
PY
