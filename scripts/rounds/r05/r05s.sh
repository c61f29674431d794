#!/bin/bash
# look-back-1 window as a constant per path (1 or 4, the plan chooses): base (auto) / w4 / w1 / f5 (auto,
# fifth image load plain)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05s; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_two_tables.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg3 base w4 w1 f5 base w4 w1 f5 base w4 w1 f5 base w4 w1 f5 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base w4 f5 base w4 f5 2>&1 | grep -v "overhang\|restart" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
for w in clipped uniform ljpeg3; do
  python scripts/exp_ab.py run --what $w base w4 f5 base w4 f5 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
