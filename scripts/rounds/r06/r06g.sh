#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06g; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_per_component_tables.py tests/test_gpu_two_tables.py -q 2>&1 | grep -v "amdgpu.ids" | tail -6 | tee $O/pytest_pt.txt
RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so timeout 300 python -m pytest tests/test_gpu_per_component_tables.py -q -x -k mixed_plan 2>&1 | grep -v "amdgpu.ids" | grep -E "K0 words|stream [0-9]:" | grep -o "K0 words.*\|stream [0-9]\|reasons 0x[0-9a-f]* (block [0-9]* slot [0-9]* symbols before it [0-9]* base [0-9]*\|flags [0-9]*" | head -12 | tee $O/mixed_why.txt
timeout 300 python bench_ljpeg.py --only ljpegpt 2>/dev/null | grep -E "ms_per_step|\"lj_|bit_exact\"" | tee $O/bench_pt.txt
