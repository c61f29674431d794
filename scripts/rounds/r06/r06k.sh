#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06k; mkdir -p $O
cd $REPO
for r in 1 2; do
echo "-- chunks / bands" | tee -a $O/chunked_timing.txt
python scripts/exp_chunked_host.py 2>&1 | grep -v amdgpu.ids | tee -a $O/chunked_timing.txt
echo "-- RSX_HOST_NO_OVERLAP=1" | tee -a $O/chunked_timing.txt
RSX_HOST_NO_OVERLAP=1 python scripts/exp_chunked_host.py 2>&1 | grep -v amdgpu.ids | tee -a $O/chunked_timing.txt
done
