#!/bin/bash
# Round 6: the whole GPU suite with the Nikon-type route and its route tests.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06p; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_nikon_routes.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -30 | tee $O/pytest_routes.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee $O/pytest_gpu.txt
