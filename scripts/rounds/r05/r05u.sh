#!/bin/bash
# look-back 1, first pass: the 64 / 32 / 16 / 8 / 4 nearest records (base = 64)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05u; mkdir -p $O
cd $REPO
RSX_LIB=$REPO/rawspeed_amd/variants/librsx_v8.so timeout 300 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py -q -x 2>&1 | tail -3 | tee $O/pytest_v8.txt
python scripts/exp_ab.py run --what cfg3 base v32 v16 v8 v4 base v32 v16 v8 v4 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base v32 v16 v8 v4 base v32 v16 v8 v4 2>&1 | grep -v "overhang\|restart\|tiles" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
for w in uniform ljpeg3; do
  python scripts/exp_ab.py run --what $w base v32 v16 v8 v4 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
