#!/bin/bash
# Round 6, fourth GPU session: the reproducer on a heap whose top is trimmed and regrown around every image
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06d; mkdir -p $O
cd $REPO
R=$REPO/scripts/repro/memcpy2d_pageable
echo "== reproducer" | tee $O/repro.txt
for cfg in "rect none heaptrim 1 40 1" "rect none heaptrim 1 40 2 prefork" "rect none heaptrim 2 30 3" "rect16 none heaptrim 1 30 4" "rows1d none heaptrim 1 30 5" "pinned none heaptrim 1 20 6"; do
  timeout 120 $R $cfg 2>&1 | grep -v "amdgpu.ids" | tail -12 | cut -c1-400 | tee -a $O/repro.txt
done
