#!/bin/bash
# last soak of the round, final library: the big-size generators (hundreds of workgroups a stream)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05zz; mkdir -p $O
cd $REPO
for m in "big 0 30" "big2 0 30" "big3 0 40" "bigdri 0 50"; do
  timeout 150 python scripts/fuzz_more.py $m 2>&1 | tail -4 | tee -a $O/fuzz_big.txt
done
