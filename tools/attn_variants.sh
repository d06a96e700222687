#!/bin/bash
# Developer: build attention.hip with extra flags and link it with the other objects of the product build as
# clip_assisted_data_labeling_amd/libclipenc_hip_<suffix>.so (same-box A/B with tools/attn_ab.py or tools/ab_multi.sh)
#   bash tools/attn_variants.sh a1="-DATTN_V=1" s0="-DATTN_STAMPS" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/clip_assisted_data_labeling_amd/csrc
make -s -j8 -C "$C"
for arg in "$@"; do
  sfx=${arg%%=*}; flags=${arg#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c "$C/attention.hip" -o "$C/build/attention_$sfx.o"
  OBJS=$(ls "$C"/build/*.o | grep -v "/attention\(_[^/]*\)\?\.o$" | grep -v "__" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/clip_assisted_data_labeling_amd/libclipenc_hip_$sfx.so" $OBJS "$C/build/attention_$sfx.o"
  echo "built $sfx ($flags)"
done
