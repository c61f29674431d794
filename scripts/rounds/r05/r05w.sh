#!/bin/bash
# results in two sets, the next run's cleared by K0 (no lj_init_results_kernel in front of a run):
# the whole GPU suite, then base against v16 (= the library two commits back)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05w; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg4 base v16 base v16 2>&1 | grep -v "overhang" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
python scripts/exp_ab.py run --what cfg3 base v16 base v16 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
