#!/bin/bash
# Round 6: the Nikon-type copy-out with two segments' table loads in flight; cfg 3 / cfg 5 as a
# check that the copy-out's new shape (lambdas) left the other instantiations where they were.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06q; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_nikon_routes.py tests/test_gpu_nikon.py tests/test_gpu_fast_path.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee $O/pytest.txt
L=$REPO/rawspeed_amd/variants/librsx_stats.so
echo "== nikon-type pixels, uncorrected=0" | tee -a $O/phases.txt
UNCORRECTED=0 RSX_DEBUG=1 RSX_LIB=$L timeout 200 python scripts/exp_nk_phases.py 2>&1 | grep -E "^\[rsx\]   |single-pass phases" | head -17 | tee -a $O/phases.txt
timeout 300 python bench_ljpeg.py --only nikon 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_nikon.txt | grep -E "^nikon" | cut -c1-330
timeout 300 python bench_ljpeg.py --only cfg3 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_cfg3.txt | grep -E "ms_per_step|mpix_per_s|lj_" | head -12
timeout 300 python bench_ljpeg.py --only cfg4 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_cfg4.txt | grep -E "ms_per_step|mpix_per_s|lj_" | head -12
