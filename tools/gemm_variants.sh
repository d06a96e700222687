#!/bin/bash
# Developer: build one GEMM source with extra flags and link it with the other objects of the product build as
# clip_assisted_data_labeling_amd/libclipenc_hip_<suffix>.so (same-box A/B with tools/ab_multi.sh / tools/gemm_fp8_ab.py)
#   bash tools/gemm_variants.sh st3="-DGEMM_STAGGER=300" ...                 (gemm_persist.hip + gemm_tri.hip, which includes it)
#   SRC=gemm_fp8 bash tools/gemm_variants.sh nozc="-DF8_LNF_ZERO_C=0" ...     (another source)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/clip_assisted_data_labeling_amd/csrc
SRC=${SRC:-gemm_persist}
FILES=$SRC; [ "$SRC" = gemm_persist ] && FILES="gemm_persist gemm_tri"
make -s -j8 -C "$C"
for arg in "$@"; do
  sfx=${arg%%=*}; flags=${arg#*=}
  VOBJ=""
  for f in $FILES; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c "$C/$f.hip" -o "$C/build/${f}__$sfx.o"
    VOBJ="$VOBJ $C/build/${f}__$sfx.o"
  done
  OBJS=$(ls "$C"/build/*.o | grep -v "__" | grep -v "/\($(echo $FILES | sed 's/ /\\|/g')\)\.o" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/clip_assisted_data_labeling_amd/libclipenc_hip_$sfx.so" $OBJS $VOBJ
  echo "built $sfx ($flags)"
done
