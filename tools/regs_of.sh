#!/bin/bash
# tools/regs_of.sh <row type: f32|f16|i8|u8> <mangled-name fragment> [extra hipcc flags]: registers / scratch / occupancy of the matching search_kernel instantiations
T=$1; PAT=$2; shift 2
cd "$(dirname "$0")/../scalablevectorsearch_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -S --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" svsx_search_$T.hip -o /tmp/asm/${T}_cur.s 2>&1 | awk -v pat="$PAT" '/Function Name/ {keep = index($0, pat) > 0} keep && /(Function Name|VGPRs:|TotalSGPRs|ScratchSize|Occupancy)/ {sub(/.*remark: +/, ""); sub(/ \[-Rpass.*/, ""); printf "%s | ", $0} keep && /Occupancy/ {print ""}'
