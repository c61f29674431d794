#!/bin/bash
# Round 6: the Nikon-type copy-out with its table loads issued at once; the route tests.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06o; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_nikon_routes.py tests/test_gpu_nikon.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -30 | tee $O/pytest.txt
L=$REPO/rawspeed_amd/variants/librsx_stats.so
for unc in 1 0; do
echo "== nikon-type pixels, uncorrected=$unc" | tee -a $O/phases.txt
UNCORRECTED=$unc RSX_DEBUG=1 RSX_LIB=$L timeout 200 python scripts/exp_nk_phases.py 2>&1 | grep -E "^\[rsx\]   |single-pass phases" | head -17 | tee -a $O/phases.txt
done
timeout 300 python bench_ljpeg.py --only nikon 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_nikon.txt | grep -E "^nikon" | cut -c1-330
