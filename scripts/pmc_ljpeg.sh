#!/bin/bash
# PMC passes over the cfg-3 LJPEG leg (run on the GPU box): instruction mix,
# wait / active cycles, LDS conflicts for the synchronisation and final-decode kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_lj
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/pl_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pl_$i -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
  cp $(find /tmp/pl_$i -name "*counter_collection.csv" | head -1) $OUT/set$i.csv
  python $REPO/scripts/pmc_summary.py $OUT/set$i.csv lj_ > $OUT/set$i.txt
  grep -A6 "lj_fast_kernel\|lj_unstuff_kernel" $OUT/set$i.txt | grep -v "^--"
done
python $REPO/scripts/pmc_ljpeg_json.py $OUT
