#!/bin/bash
# look-back-1 window per plan + probe launches: base (new) / w4 (new, window of 256) / lb1 (old head,
# window of 64) / r5a (shipped before); correctness of the new library; first-run traffic
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05q; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_ljpeg.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg3 base w4 r5a lb1 base w4 r5a lb1 base w4 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base w4 r5a base w4 r5a > $O/ab_cfg4.txt 2>&1
cat $O/ab_cfg4.txt
for w in clipped uniform ljpeg3; do
  python scripts/exp_ab.py run --what $w base w4 r5a base w4 r5a 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
bash scripts/pmc_ljpeg_traffic.sh > $O/traffic.txt 2>&1
tail -12 $O/traffic.txt
mkdir -p $O/ljpeg_traffic; cp gpurun_out/pmc_lj_traffic/* $O/ljpeg_traffic/
python - <<'PY' | tee $O/first_run_fetch.txt
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/pmc_lj_traffic/ljpeg_pmc_FETCH_SIZE.csv')) if 'lj_fast' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Dispatch_Id']))
print("FETCH_SIZE (KB) of the single-pass kernel's launches in order:", [round(float(r['Counter_Value'])) for r in rows])
PY
