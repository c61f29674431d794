#!/bin/bash
# Round 6: the soak's second failure (RSX_FUZZ_BASE=155, ragged seed 18: consumed bytes of two tiles 5 and 4
# instead of 11357 and 12149, status OK): does it repeat?
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06v; mkdir -p $O
cd $REPO
for k in 1 2 3 4; do
  RSX_FUZZ_BASE=155 timeout 600 python -m pytest "tests/test_gpu_host_path_ragged.py::test_ragged_rectangles_come_back_whole_from_six_threads" -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-1200 | tee -a $O/again.txt
done
for k in 1 2 3; do
  RSX_FUZZ_BASE=155 timeout 900 python -m pytest tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_per_component_tables.py tests/test_gpu_nikon_routes.py tests/test_gpu_host_path_ragged.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-1200 | tee -a $O/again.txt
done
