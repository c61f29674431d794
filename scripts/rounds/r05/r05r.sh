#!/bin/bash
# the first run's instantiation (PROBE) + look-back-1 window per plan: base (new) / w4 (new, window of
# 256) / r5a (shipped before) / lb1 (before, window of 64); then the fuzz soak (scripts/rounds/r05/r05o.sh)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05r; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_ljpeg.py tests/test_gpu_two_tables.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
python scripts/exp_ab.py run --what cfg3 base w4 r5a lb1 base w4 r5a lb1 base w4 r5a lb1 base w4 r5a lb1 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base r5a base r5a 2>&1 | grep -v "two_tables\|overhang\|restart" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
for w in clipped uniform; do
  python scripts/exp_ab.py run --what $w base r5a base r5a 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
python - <<'PY' | tee $O/first_run_fetch.txt
import csv, glob
f = glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'lj_fast' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Dispatch_Id']))
print("FETCH_SIZE (KB) of the single-pass kernel's launches in order:", [(r['Kernel_Name'][r['Kernel_Name'].find('lj_fast'):][:34], round(float(r['Counter_Value']))) for r in rows])
PY
cd $REPO
bash scripts/rounds/r05/r05o.sh 2>&1 | tail -16
