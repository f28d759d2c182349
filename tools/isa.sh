#!/bin/bash
# compile one kernel file for gfx950, keep the ISA under /tmp/isa, print resource usage and the s_waitcnt histogram
set -e
F=${1:?file}
mkdir -p /tmp/isa
cd /root/repo/crane_amd/csrc
hipcc --offload-arch=gfx950 -O3 $ISA_EXTRA -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -x hip -c $F -o /tmp/isa/$(basename $F).o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|Function Name|VGPRs:|SGPRs:|Scratch|Occupancy" || true
S=/tmp/isa/$(basename $F .hip)-hip-amdgcn-amd-amdhsa-gfx950.s
echo "ISA: $S"
grep "s_waitcnt" $S | awk '{print $1,$2,$3}' | sort | uniq -c | sort -rn | head -20
echo "flat ops: $(grep -c 'flat_' $S)"
