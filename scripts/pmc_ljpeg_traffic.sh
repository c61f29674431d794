#!/bin/bash
# HBM traffic of the cfg-3 LJPEG pipeline per kernel (run on the GPU box): two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench_ljpeg.py --only cfg3,
# summarised by scripts/pmc_ljpeg_traffic.py into profiles/rNN/ljpeg_traffic.json.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_lj_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/plt_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/plt_$c -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
  cp $(find /tmp/plt_$c -name "*counter_collection.csv" | head -1) $OUT/ljpeg_pmc_$c.csv
done
python $REPO/scripts/pmc_ljpeg_traffic.py $OUT
