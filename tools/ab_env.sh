#!/bin/bash
# Developer: interleaved bench.py runs of the DIAGNOSTIC library with and without one of its environment switches on one box.
#   bash tools/ab_env.sh <pairs> <dtype> CLIPENC_CLS_KV=1
R=$GRAFT_REPO_ROOT; N=${1:-3}; DT=${2:-bf16}; SW=$3
export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_diag.so
for i in $(seq $N); do
  for v in off on; do
    if [ $v = on ]; then E="env $SW"; else E="env"; fi
    timeout -k 10 200 $E python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --dtype $DT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$SW $v', d['value'], {a.replace('gemm_persist_kernel','g').replace('gemm_fp8_kernel','f8').replace('attn_stream_kernel','attn'): round(b,2) for a,b in k.items() if b>5 or 'attn' in a})"
  done
done
