#!/bin/bash
# Round profile collection (run on the GPU box through gpurun): bench JSON, kernel
# stats of the same command under rocprofv3, and the two PMC passes of the headline
# kernel.  Outputs land in gpurun_out/prof_r01/ (copied to profiles/r01/ afterwards).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_full.json 2> $OUT/bench_full.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- \
  python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- \
    python $REPO/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-cfg5 > /dev/null 2>&1
  cp $(find /tmp/p_$c -name "*counter_collection.csv" | head -1) $OUT/unpack_pmc_$c.csv
done
ls -la $OUT
tail -c 600 $OUT/bench_full.json
