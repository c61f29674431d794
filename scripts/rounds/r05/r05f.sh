#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05f; mkdir -p $O
cd $REPO
S="8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26"
echo "== poisoned"; POISON_GB=6 RSX_DEBUG=1 timeout 200 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 | tee $O/diag_poison.txt | grep -v "^\[rsx\]" | head -60
