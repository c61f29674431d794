#!/bin/bash
# phase statistics of the experiment build at the final kernels (the collection ran with a stale one)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r05; mkdir -p $OUT
cd $REPO
RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
  python $REPO/scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-400 > $OUT/cfg3_phase_and_round_stats.txt
WHAT=cfg4mt RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
  python $REPO/scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-400 > $OUT/cfg4_two_tables_phase_and_round_stats.txt
WHAT=cfg4 RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
  python $REPO/scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-400 > $OUT/cfg4_phase_and_round_stats.txt
head -20 $OUT/cfg3_phase_and_round_stats.txt
head -20 $OUT/cfg4_phase_and_round_stats.txt
bash scripts/rounds/r05/r05o.sh 2>&1 | tail -14
