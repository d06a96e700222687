#!/bin/bash
# Developer: interleaved bench.py runs of the current library (cur) and libclipenc_hip_v1.so (v1) on one box
R=$GRAFT_REPO_ROOT; N=${1:-3}; DT=${2:-bf16}
for i in $(seq $N); do
  for v in cur v1; do
    if [ $v = v1 ]; then export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_v1.so; else unset CLIPENC_LIB_PATH; fi
    timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --dtype $DT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$v', d['value'], {a.replace('gemm_persist_kernel','g').replace('gemm_fp8_kernel','f8'): round(b,1) for a,b in k.items() if b>5})"
  done
done
