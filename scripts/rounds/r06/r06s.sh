#!/bin/bash
# Round 6: fuzz soak of the round's routes -- other seeds for the fuzz tests of the single-pass kernel,
# tables per component, two tables, ragged host downloads, the Nikon-type plans.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06s; mkdir -p $O
cd $REPO
for base in 101 102 103 104 105 106; do
  echo "== RSX_FUZZ_BASE=$base" | tee -a $O/soak.txt
  RSX_FUZZ_BASE=$base timeout 900 python -m pytest tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_per_component_tables.py tests/test_gpu_nikon_routes.py tests/test_gpu_host_path_ragged.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/soak.txt
done
