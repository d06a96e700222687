#!/bin/bash
# Developer: sample board power / clocks with rocm-smi while the bench runs (run through gpurun)
R=$GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor junction\)|hotspot" | tr '\n' ' '; echo; sleep 0.5; done) > $R/gpurun_out/power_samples.txt &
SMI=$!
python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_power.json 2>/dev/null
wait $SMI
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
sed -n '5,40p' $R/gpurun_out/power_samples.txt | cut -c1-260 | awk 'NR%3==0'
