#!/bin/bash
# Round 6, sixth GPU session: table-per-phase streams on the single-pass kernel; the scalar-cache-invalidating
# instantiations of device-laid-out plans; why a run of the two-context test was redone by the slow pipeline.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06f; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_per_component_tables.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -25 | tee $O/pytest_pt.txt
timeout 900 python -m pytest tests/test_gpu_two_tables.py tests/test_gpu_fast_path.py tests/test_gpu_fuzz_r05.py tests/test_gpu_fast_fuzz.py tests/test_gpu_ljpeg.py -q 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee $O/pytest_lj.txt
for r in 1 2 3; do
RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so timeout 300 python -m pytest tests/test_gpu_two_contexts.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | grep -E "reasons 0x[1-9a-f]|redone|passed|failed" | cut -c1-500 | tee -a $O/two_ctx_why.txt
done
