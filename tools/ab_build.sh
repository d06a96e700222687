#!/bin/bash
# Developer: build the library of another git revision as libclipenc_hip_v1.so for same-box A/B runs
#   bash tools/ab_build.sh [rev]     then on the GPU box:  CLIPENC_LIB_PATH=$GRAFT_REPO_ROOT/clip_assisted_data_labeling_amd/libclipenc_hip_v1.so python bench.py ...
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$ROOT" archive "$REV" clip_assisted_data_labeling_amd/csrc include | tar -x -C "$T"
make -s -j8 -C "$T/clip_assisted_data_labeling_amd/csrc" 2>&1 | grep -i error || true
cp "$T/clip_assisted_data_labeling_amd/libclipenc_hip.so" "$ROOT/clip_assisted_data_labeling_amd/libclipenc_hip_v1.so"
rm -rf "$T"
echo "built $REV -> clip_assisted_data_labeling_amd/libclipenc_hip_v1.so"
