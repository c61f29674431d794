#!/bin/bash
# does the big3 failure of scripts/rounds/r05/r05zz.sh come back behind OTHER processes' work, and does the
# first-run scalar-cache invalidation change it?  noinv = the library without it
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05g; mkdir -p $O
cd $REPO
S="0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32 33 34 35 36 37 38 39"
timeout 100 python scripts/fuzz_more.py big 0 30 2>&1 | tail -1
timeout 100 python scripts/fuzz_more.py big2 0 30 2>&1 | tail -1
echo "== noinv"; RSX_LIB=$REPO/rawspeed_amd/variants/librsx_noinv.so timeout 120 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | cut -c1-400 | tee $O/diag_noinv.txt
timeout 100 python scripts/fuzz_more.py big 0 30 2>&1 | tail -1
timeout 100 python scripts/fuzz_more.py big2 0 30 2>&1 | tail -1
echo "== base (invalidates)"; timeout 120 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | cut -c1-400 | tee $O/diag_base.txt
