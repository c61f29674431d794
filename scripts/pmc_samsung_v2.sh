#!/bin/bash
# PMC passes over the SamsungV2 leg (run on the GPU box): instruction mix, wait / active
# cycles, LDS conflicts of its kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sv2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf /tmp/ps_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/ps_$i -- \
    python $REPO/bench_ljpeg.py --only samsung_v2 --steps 1 --no-cpu > /dev/null 2>&1
  cp $(find /tmp/ps_$i -name "*counter_collection.csv" | head -1) $OUT/set$i.csv
  python $REPO/scripts/pmc_summary.py $OUT/set$i.csv sv2_ > $OUT/set$i.txt
  grep -A6 "sv2_recon_kernel" $OUT/set$i.txt | grep -v "^--"
done
