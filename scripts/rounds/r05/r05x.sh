#!/bin/bash
# lj_scan_kernel's tail pre-loaded in front of the scan (base) against the tail as it was (prev)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05x; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg4 base prev base prev 2>&1 | grep -v "overhang" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
python scripts/exp_ab.py run --what cfg3 base prev base prev > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
