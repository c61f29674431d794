#!/bin/bash
# lj_scan_kernel: a single-pass stream's first-pass scan with every load asked for at once
# (base) against the loop (v16 = the same single-pass kernel, the scan as it was)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05v; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests -q -x -m gpu -k "ljpeg or fast or two_tables or fuzz or cr2 or dng or baseline" 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg3 base v16 base v16 base v16 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base v16 base v16 2>&1 | grep -v "overhang" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
for w in uniform ljpeg3 clipped; do
  python scripts/exp_ab.py run --what $w base v16 base v16 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
