#!/bin/bash
# Round 6: the six-thread ragged soak on the library that checks, in every LJPEG host call, the device
# input against the caller's and a second run of the plan against the first (-DRSX_DIAG_VERIFY).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06x; mkdir -p $O
cd $REPO
for base in $(seq 300 399); do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_verify.so RSX_FUZZ_BASE=$base timeout 300 python scripts/soak_ragged.py 2>&1 | grep -v "amdgpu.ids" | grep -E "EVENT|again|soak base|RSX_DIAG_VERIFY: [IaA]|Error|error" | cut -c1-1500 >> $O/soak.txt
done
grep -c "soak base" $O/soak.txt; grep -c EVENT $O/soak.txt; grep -c "DIAG_VERIFY" $O/soak.txt; grep -B2 -A4 "DIAG_VERIFY\|EVENT" $O/soak.txt | head -60
