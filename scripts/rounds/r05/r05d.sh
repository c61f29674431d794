#!/bin/bash
# diagnose the big3 seeds that failed (scripts/rounds/r05/r05zz.sh): shipped library, then older variants
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05d; mkdir -p $O
cd $REPO
timeout 120 python scripts/fuzz_diag.py big3 10 12 14 22 2>&1 | grep -v amdgpu.ids | tee $O/diag_base.txt
for v in w4 r5a; do
  echo "== $v"; RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so timeout 100 python scripts/fuzz_more.py big3 8 18 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/diag_variants.txt
done
