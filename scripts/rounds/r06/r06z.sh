#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06z; mkdir -p $O
cd $REPO
echo "== fixed" | tee $O/regression.txt
timeout 600 python -m pytest "tests/test_gpu_host_path_ragged.py::test_plan_creation_against_a_busy_null_stream" -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-600 | tee -a $O/regression.txt
echo "== as it was (-DRSX_NO_CREATE_SYNC)" | tee -a $O/regression.txt
RSX_LIB=$REPO/rawspeed_amd/variants/librsx_nosync.so timeout 600 python -m pytest "tests/test_gpu_host_path_ragged.py::test_plan_creation_against_a_busy_null_stream" -m gpu -q 2>&1 | grep -v "amdgpu.ids" | grep -E "^E  |passed|failed" | cut -c1-900 | tee -a $O/regression.txt
