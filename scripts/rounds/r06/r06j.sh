#!/bin/bash
# A/B: the pair-window decode (one window read for two symbols) against the shipped kernel, interleaved
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06j; mkdir -p $O
cd $REPO
for r in 1 2 3; do
  python scripts/exp_ab.py run --what cfg3 base pairwin 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_pairwin_cfg3.txt
done
for r in 1 2; do
  python scripts/exp_ab.py run --what cfg4 base pairwin 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/ab_pairwin_cfg4.txt
done
RSX_LIB=$REPO/rawspeed_amd/variants/librsx_pairwin.so timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_baseline_parity.py -q 2>&1 | tail -3 | tee $O/pytest_pairwin.txt
