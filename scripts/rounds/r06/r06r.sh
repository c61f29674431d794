#!/bin/bash
# Round 6: SamsungV1 (an explicit 10-bit table) on the Nikon-type instantiation.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06r; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_nikon.py tests/test_gpu_nikon_routes.py tests/test_gpu_raw_files.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee $O/pytest.txt
timeout 300 python bench_ljpeg.py --only samsung_v1 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_samsung_v1.txt | grep -E "mpix_per_s|ms_per_step|lj_|legacy|bit_exact\"" | head
