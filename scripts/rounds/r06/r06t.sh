#!/bin/bash
# Round 6: the one failure of the soak (RSX_FUZZ_BASE=105, ragged rectangles from six threads, in a process
# that had run the other fuzz files first): the same command again, with the message.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06t; mkdir -p $O
cd $REPO
for base in 105 105 105 105 107 108 109 110 111 112; do
  echo "== RSX_FUZZ_BASE=$base" | tee -a $O/again.txt
  RSX_FUZZ_BASE=$base timeout 900 python -m pytest tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_per_component_tables.py tests/test_gpu_nikon_routes.py tests/test_gpu_host_path_ragged.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-900 | tee -a $O/again.txt
done
