#!/bin/bash
# Developer: build gemm_persist.hip (+ gemm_tri.hip, which includes it) with extra flags and link with the other objects of the
# product build as clip_assisted_data_labeling_amd/libclipenc_hip_<suffix>.so (same-box A/B with tools/ab_multi.sh)
#   bash tools/gemm_variants.sh st3="-DGEMM_STAGGER=300" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/clip_assisted_data_labeling_amd/csrc
make -s -j8 -C "$C"
for arg in "$@"; do
  sfx=${arg%%=*}; flags=${arg#*=}
  for f in gemm_persist gemm_tri; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c "$C/$f.hip" -o "$C/build/${f}__$sfx.o"
  done
  OBJS=$(ls "$C"/build/*.o | grep -v "gemm_persist\|gemm_tri\|__" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/clip_assisted_data_labeling_amd/libclipenc_hip_$sfx.so" $OBJS "$C/build/gemm_persist__$sfx.o" "$C/build/gemm_tri__$sfx.o"
  echo "built $sfx ($flags)"
done
