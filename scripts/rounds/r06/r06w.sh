#!/bin/bash
# Round 6: many processes of the six-thread ragged case alone, with a post-mortem on any difference.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06w; mkdir -p $O
cd $REPO
for base in $(seq 200 279); do
  RSX_FUZZ_BASE=$base timeout 300 python scripts/soak_ragged.py 2>&1 | grep -v "amdgpu.ids" | grep -E "EVENT|again|soak base|Error|error" | cut -c1-1500 >> $O/soak.txt
done
grep -c "soak base" $O/soak.txt; grep -c EVENT $O/soak.txt; grep -A3 EVENT $O/soak.txt | head -40
