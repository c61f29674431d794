#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06h; mkdir -p $O
cd $REPO
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -6 | tee $O/pytest_gpu.txt
for m in "two_tables 24 124" "single 48 148"; do timeout 300 python scripts/fuzz_more.py $m 2>&1 | tail -2 | tee -a $O/fuzz.txt; done
for b in 1 2 3; do RSX_FUZZ_BASE=$b timeout 300 python -m pytest tests/test_gpu_per_component_tables.py tests/test_gpu_fuzz_r05.py -q 2>&1 | tail -2 | tee -a $O/fuzz.txt; done
timeout 300 python bench_ljpeg.py --only cfg4mt 2>/dev/null | grep -E "ms_per_step" | head -4 | tee $O/cfg4mt.txt
timeout 300 python bench_ljpeg.py --only cfg3 2>/dev/null | grep -E "ms_per_step|\"lj_" | head -6 | tee $O/cfg3.txt
