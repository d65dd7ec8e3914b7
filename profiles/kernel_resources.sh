#!/bin/bash
# Prints vgpr / sgpr / LDS / scratch / occupancy of every kernel of a .hip source compiled for gfx950 (no GPU needed).
# usage: profiles/kernel_resources.sh vg-renderer_amd/csrc/vgx_stroke.hip [extra flags]
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -c $SRC -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | awk '
/Function Name:/ {name=$NF}
/ VGPRs:/ {v=$NF}
/TotalSGPRs:/ {s=$NF}
/ScratchSize/ {p=$NF}
/Occupancy/ {o=$NF}
/VGPRs Spill/ {sp=$NF}
/LDS Size/ {printf "%-60s vgpr=%s sgpr=%s scratch=%s spill=%s occ=%s lds=%s\n", substr(name,1,60), v, s, p, sp, o, $NF}'
