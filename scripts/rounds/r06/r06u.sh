#!/bin/bash
# Round 6: a longer soak of the fuzz files in one process each (the soak's one failure -- ragged
# rectangles from six threads -- did not repeat in 16 further processes): 36 processes, the message kept.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06u; mkdir -p $O
cd $REPO
for base in $(seq 120 155); do
  echo "== RSX_FUZZ_BASE=$base" >> $O/soak.txt
  RSX_FUZZ_BASE=$base timeout 900 python -m pytest tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_per_component_tables.py tests/test_gpu_nikon_routes.py tests/test_gpu_host_path_ragged.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-1200 >> $O/soak.txt
done
grep -c passed $O/soak.txt; grep -B2 -A6 "failed" $O/soak.txt | head -60
