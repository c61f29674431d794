#!/bin/bash
# Round 6, third GPU session: the reproducer behind a fork (copy-on-write heap) and on the brk heap; the
# round-5 recipe on the diagnostic library in SIX processes at a time (one process in five shows the defect),
# with the extended checks: which host pages, what the device reads there, repeat / CPU touch / page-lock /
# one contiguous copy.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06c; mkdir -p $O
cd $REPO
R=$REPO/scripts/repro/memcpy2d_pageable
echo "== reproducer" | tee $O/repro.txt
for cfg in "rect none heap 1 30 1 prefork" "rect none heap 4 30 2 prefork" "rect none heap 2 30 3" "rect fork heap 1 30 4 prefork"; do
  timeout 120 $R $cfg 2>&1 | grep -v "amdgpu.ids" | tail -6 | cut -c1-400 | tee -a $O/repro.txt
done
echo "== library (diag2d), 3 rounds of 6 processes side by side" | tee $O/diag.txt
for round in 1 2 3; do
  for k in 1 2 3 4 5 6; do
    ( RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 300 python scripts/fuzz_more.py big3 0 40 2>&1 \
        | grep -v "amdgpu.ids" | grep "FAILED\|RSX_DIAG\|failed:" | cut -c1-600 > $O/diag_${round}_$k.txt ) &
  done
  wait
  for k in 1 2 3 4 5 6; do echo "--- round $round process $k"; cat $O/diag_${round}_$k.txt; done | tee -a $O/diag.txt | grep -v "rect .* row .*host != device" | tail -60
done
