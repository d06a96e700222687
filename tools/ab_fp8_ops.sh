#!/bin/bash
# Developer: tools/gemm_fp8_ab.py once per library variant, interleaved, on one box.
#   bash tools/ab_fp8_ops.sh <rounds> "<AB_ONLY list>" <suffix> [<suffix> ...]     suffix "cur" = libclipenc_hip.so
R=$GRAFT_REPO_ROOT; N=${1:-2}; export AB_ONLY=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    if [ $v = cur ]; then unset CLIPENC_LIB_PATH; else export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_$v.so; fi
    timeout -k 10 300 python $R/tools/gemm_fp8_ab.py 2>&1 | grep -v Warning
  done
done
