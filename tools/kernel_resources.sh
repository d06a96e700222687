#!/bin/bash
# Developer: per-kernel register / spill usage of one HIP source as hipcc reports it, and how often the ISA writes M0
# outside this repo's own `s_mov_b32 m0` LDS-DMA statements (no GPU needed).
#   bash tools/kernel_resources.sh clip_assisted_data_labeling_amd/csrc/gemm_persist.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
            print(name[:64].ljust(64), "vgpr", cur.get("VGPRs"), "agpr", cur.get("AGPRs"), "sgpr", cur.get("TotalSGPRs"),
                  "spill v/s", cur.get("VGPRs Spill"), cur.get("SGPRs Spill"), "scratch", cur.get("ScratchSize [bytes/lane]"),
                  "occ", cur.get("Occupancy [waves/SIMD]"))
            cur = None
'
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S "$f" -o /tmp/_kr.s "$@" 2>/dev/null
echo "M0 writes in the ISA: $(grep -c 'm0' /tmp/_kr.s) of which 's_mov_b32 m0' $(grep -c 's_mov_b32 m0' /tmp/_kr.s)"
