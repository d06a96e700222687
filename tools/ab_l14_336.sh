#!/bin/bash
# Developer: interleaved tools/bench_model.py runs (ViT-L-14-336, 480 crops) of several library builds on one box.
#   bash tools/ab_l14_336.sh <pairs> <suffix> [<suffix> ...]     suffix "cur" = libclipenc_hip.so
R=$GRAFT_REPO_ROOT; N=${1:-3}; shift 1
for i in $(seq $N); do
  for v in "$@"; do
    if [ $v = cur ]; then unset CLIPENC_LIB_PATH; else export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_$v.so; fi
    echo "== $v"; timeout -k 10 300 python $R/tools/bench_model.py --arch ViT-L-14-336 --crops 480 2>&1 | grep -v "Warn\|amdgpu.ids"
  done
done
