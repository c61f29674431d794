#!/bin/bash
# Round 6: the regression test of the creation-time memsets on the fixed library and on the library as it
# was; then the whole GPU suite; then the multi-process soak again on the fixed library.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06y; mkdir -p $O
cd $REPO
echo "== fixed" | tee $O/regression.txt
timeout 600 python -m pytest "tests/test_gpu_host_path_ragged.py::test_plan_creation_against_a_busy_null_stream" -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-600 | tee -a $O/regression.txt
echo "== as it was (-DRSX_NO_CREATE_SYNC)" | tee -a $O/regression.txt
RSX_LIB=$REPO/rawspeed_amd/variants/librsx_nosync.so timeout 600 python -m pytest "tests/test_gpu_host_path_ragged.py::test_plan_creation_against_a_busy_null_stream" -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-900 | tee -a $O/regression.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee $O/pytest_gpu.txt
for base in $(seq 400 439); do
  RSX_FUZZ_BASE=$base timeout 300 python scripts/soak_ragged.py 2>&1 | grep -v "amdgpu.ids" | grep -E "EVENT|again|soak base|Error|error" | cut -c1-1500 >> $O/soak.txt
done
grep -c "soak base" $O/soak.txt; grep -c EVENT $O/soak.txt
