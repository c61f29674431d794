#!/bin/bash
# Round 6, fifth GPU session: the shipped library with download_rects at every download -- the new host-path
# and two-context tests, the whole suite, the big3 soak, the host-path timings; the diagnostic library once more
# in sequence (the defect shows in about one process in ten).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06e; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_host_path_ragged.py tests/test_gpu_two_contexts.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
for r in 1 2; do timeout 150 python scripts/fuzz_more.py big3 0 40 2>&1 | tail -1 | tee -a $O/big3_shipped.txt; done
timeout 100 python scripts/fuzz_more.py bigdri 0 30 2>&1 | tail -1 | tee -a $O/big3_shipped.txt
timeout 300 python bench_ljpeg.py --only host 2> $O/host.err | tail -c 3000 | tee $O/host_path.json
for r in 1 2 3 4 5; do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 150 python scripts/fuzz_more.py big3 0 40 2>&1 | grep -v "amdgpu.ids" | grep "RSX_DIAG\|failed:" | cut -c1-600 | tee -a $O/diag.txt | grep -v "host != device"
done
