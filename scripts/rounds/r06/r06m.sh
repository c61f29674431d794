#!/bin/bash
# Round 6: Nikon-type pixels on the single-pass kernel (fast_nk) -- the Nikon / Pentax tests, the
# whole-file tests, the legs' timings with the kernel table.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06m; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_nikon.py tests/test_gpu_raw_files.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -25 | tee $O/pytest_nikon.txt
timeout 300 python bench_ljpeg.py --only nikon 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_nikon.txt | cut -c1-900
timeout 300 python bench_ljpeg.py --only pentax 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_pentax.txt | cut -c1-900
