#!/bin/bash
# soak of the round-5 differential fuzzers: the committed seeds, then 12 more seed bases
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05o
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_fuzz_r05.py -q -x 2>&1 | tail -15 > $OUT/fuzz_base0.txt
cat $OUT/fuzz_base0.txt
for b in 1 2 3 4 5 6 7 8 9 10 11 12; do
  RSX_FUZZ_BASE=$b timeout 300 python -m pytest tests/test_gpu_fuzz_r05.py tests/test_gpu_fast_fuzz.py -q 2>&1 | tail -12 > $OUT/fuzz_base$b.txt
  echo "base $b: $(tail -1 $OUT/fuzz_base$b.txt)"
done
