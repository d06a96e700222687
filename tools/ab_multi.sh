#!/bin/bash
# Developer: interleaved bench.py runs of several library builds on one box.
#   bash tools/ab_multi.sh <pairs> <dtype> <suffix> [<suffix> ...]     suffix "cur" = libclipenc_hip.so, else libclipenc_hip_<suffix>.so
R=$GRAFT_REPO_ROOT; N=${1:-3}; DT=${2:-bf16}; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    if [ $v = cur ]; then unset CLIPENC_LIB_PATH; else export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_$v.so; fi
    timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --dtype $DT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$v', d['value'], {a.replace('gemm_persist_kernel','g').replace('gemm_fp8_kernel','f8').replace('attn_stream_kernel','attn'): round(b,1) for a,b in k.items() if b>5})"
  done
done
