#!/bin/bash
# Round 6, last: the fuzz files in one process each on the final library, other seeds.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06last; mkdir -p $O
cd $REPO
for base in $(seq 500 513); do
  echo "== RSX_FUZZ_BASE=$base" >> $O/soak.txt
  RSX_FUZZ_BASE=$base timeout 300 python -m pytest tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_per_component_tables.py tests/test_gpu_nikon_routes.py tests/test_gpu_host_path_ragged.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-900 >> $O/soak.txt
done
grep -c passed $O/soak.txt; grep -c failed $O/soak.txt; grep -B1 -A4 failed $O/soak.txt | head -20
