#!/bin/bash
# where do the single-pass kernel's extra HBM reads come from (394 MB against 320 MB in round 4)?
# FETCH_SIZE of cfg 3 per kernel for: shipped, old ticket head, look-back-1 window of 64 / 128 records
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA_RDREQ[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_HIT[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*" | sort -u | head -40 > $O/counters.txt
cat $O/counters.txt
summ() { python - "$1" "$2" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]; n = n[n.find("lj_"):].split("(")[0] if "lj_" in n else n[:30]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(acc):
    if n.startswith("lj_fast") or n.startswith("lj_unstuff"):
        print(sys.argv[2], n, {c: round(sum(v) / len(v), 1) for c, v in acc[n].items()}, "launches", len(next(iter(acc[n].values()))))
PY
}
for v in base tick lb1 lb2; do
  if [ $v != base ]; then export RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so; else unset RSX_LIB; fi
  rm -rf /tmp/pp_$v
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pp_$v -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
  f=$(find /tmp/pp_$v -name "*counter_collection.csv" | head -1)
  cp $f $O/fetch_$v.csv
  summ $f "$v" | tee -a $O/fetch_summary.txt
done
for v in base lb1; do
  if [ $v != base ]; then export RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so; else unset RSX_LIB; fi
  rm -rf /tmp/pq_$v
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/pq_$v -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > $O/rdreq_$v.log 2>&1
  f=$(find /tmp/pq_$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && summ $f "$v" | tee -a $O/rdreq_summary.txt
done
unset RSX_LIB
cd $REPO
python scripts/exp_ab.py run --what cfg3 base lb1 lb2 tick base lb1 lb2 tick > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base lb1 lb2 base lb1 lb2 2>&1 | grep -v "^\S* *  " > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
timeout 300 env RSX_LIB=$REPO/rawspeed_amd/variants/librsx_lb1.so python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py -q -x 2>&1 | tail -3 | tee $O/pytest_lb1.txt
