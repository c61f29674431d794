#!/bin/bash
# the big3 soak fails on some seeds in some runs: repeat with details (shipped), and on the 256-record look-back (w4)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05e; mkdir -p $O
cd $REPO
S="0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32 33 34 35 36 37 38 39"
for r in 1 2 3; do
  echo "== base run $r"; timeout 120 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | tee -a $O/diag_base.txt
done
for r in 1 2; do
  echo "== w4 run $r"; RSX_LIB=$REPO/rawspeed_amd/variants/librsx_w4.so timeout 120 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | tee -a $O/diag_w4.txt
done
