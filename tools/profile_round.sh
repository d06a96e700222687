#!/bin/bash
# Developer: collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root).
#   bash tools/profile_round.sh r01 [bf16|fp8] [ViT-L-14|ViT-L-14-336]
set -u
TAG=${1:-r01}
DT=${2:-bf16}
MODEL=${3:-ViT-L-14}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --dtype $DT --model $MODEL > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-power-ceiling --secondary dedup --dtype $DT --model $MODEL > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-ceiling --secondary dedup --dtype $DT --model $MODEL > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-ceiling --secondary dedup --dtype $DT --model $MODEL > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-ceiling --secondary dedup --dtype $DT --model $MODEL > $OUT/pmc_sq.log 2>&1
ls $OUT
