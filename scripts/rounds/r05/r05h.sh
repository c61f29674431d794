#!/bin/bash
# ragged tile sets come back through one contiguous copy: the host-path tests, then the big3 soak
# behind other processes' work, twice
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05h; mkdir -p $O
cd $REPO
S="0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32 33 34 35 36 37 38 39"
timeout 200 python -m pytest tests/test_gpu_fuzz_r05.py tests/test_gpu_fast_fuzz.py tests/test_gpu_two_tables.py tests/test_gpu_dropin.py -q -x 2>&1 | tail -2 | tee $O/pytest.txt
timeout 60 python scripts/fuzz_more.py big 0 20 2>&1 | tail -1
timeout 60 python scripts/fuzz_more.py big2 0 20 2>&1 | tail -1
for r in 1 2; do
  echo "== big3 run $r"; timeout 100 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | cut -c1-300 | tee -a $O/diag_big3.txt
done
echo "== bigdri"; timeout 60 python scripts/fuzz_more.py bigdri 0 30 2>&1 | tail -1 | tee $O/bigdri.txt
