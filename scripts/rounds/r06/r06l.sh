#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06l; mkdir -p $O
cd $REPO
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee $O/pytest_gpu.txt
