#!/bin/bash
# Round profile collection (run on the GPU box through gpurun):
#   bench JSON; kernel stats of the same command under rocprofv3; the two PMC passes of
#   the headline kernel; kernel stats + per-kernel HBM traffic of the LJPEG legs; the
#   re-decode round statistics of an experiment build.
# Outputs land in gpurun_out/prof_$ROUND/ (copied to profiles/$ROUND/ afterwards).
set -u
ROUND=${ROUND:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_full.json 2> $OUT/bench_full.log
rm -rf /tmp/p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- \
  python $REPO/bench.py --no-cpu-baseline --no-extra --no-cfg5 > $OUT/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- \
    python $REPO/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-cfg5 > /dev/null 2>&1
  cp $(find /tmp/p_$c -name "*counter_collection.csv" | head -1) $OUT/unpack_pmc_$c.csv
done
python $REPO/scripts/pmc_to_json.py $OUT > /dev/null
for leg in cfg3 cfg4; do
  rm -rf /tmp/p_$leg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$leg -- \
    python $REPO/bench_ljpeg.py --only $leg --steps 10 --no-cpu > $OUT/${leg}_under_rocprof.json 2> /dev/null
  cp $(find /tmp/p_$leg -name "*kernel_stats.csv" | head -1) $OUT/${leg}_kernel_stats.csv
done
# the Nikon-type instantiation (round 6): kernel stats of the Nikon leg, both output modes
rm -rf /tmp/p_nikon
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nikon -- \
  python $REPO/bench_ljpeg.py --only nikon --steps 10 --no-cpu > $OUT/nikon_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_nikon -name "*kernel_stats.csv" | head -1) $OUT/nikon_kernel_stats.csv
bash $REPO/scripts/pmc_ljpeg_traffic.sh > /dev/null 2>&1
mkdir -p $OUT/ljpeg_traffic
cp $REPO/gpurun_out/pmc_lj_traffic/* $OUT/ljpeg_traffic/
# instruction mix of the LJPEG kernels (cfg 3): needs cfg3_kernel_stats.csv (above) for the times
mkdir -p $REPO/gpurun_out/pmc_lj; cp $OUT/cfg3_kernel_stats.csv $REPO/gpurun_out/pmc_lj/ 2>/dev/null
bash $REPO/scripts/pmc_ljpeg.sh > $OUT/ljpeg_pmc.log 2>&1
mkdir -p $OUT/ljpeg_pmc
cp $REPO/gpurun_out/pmc_lj/*.txt $REPO/gpurun_out/pmc_lj/ljpeg_pmc.json $OUT/ljpeg_pmc/ 2>/dev/null
cp $REPO/bench_extra.json $OUT/ 2>/dev/null
# re-decode round statistics (experiment build: RSX_EXPERIMENT collects them, RSX_DEBUG prints)
if [ -f $REPO/rawspeed_amd/variants/librsx_stats.so ]; then
  RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
    python $REPO/scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-400 > $OUT/cfg3_phase_and_round_stats.txt
  WHAT=cfg4mt RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
    python $REPO/scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-400 > $OUT/cfg4_two_tables_phase_and_round_stats.txt
fi
if [ -f $REPO/rawspeed_amd/variants/librsx_stats.so ]; then
  for unc in 0 1; do
    echo "== Nikon 6016x4016 x 8, uncorrected=$unc"
    UNCORRECTED=$unc RSX_DEBUG=1 RSX_LIB=$REPO/rawspeed_amd/variants/librsx_stats.so \
      python $REPO/scripts/exp_nk_phases.py 2>&1 | grep "^\[rsx\]  \|phases over" | cut -c1-200
  done > $OUT/nikon_type_phase_stats.txt
fi
python $REPO/scripts/ljpeg_limiter.py $OUT > /dev/null 2>&1
ls -la $OUT
tail -c 400 $OUT/bench_full.json
timeout 900 python -m pytest $REPO/tests -q -m gpu 2>&1 | tail -5 > $OUT/pytest_gpu_tail.txt
cat $OUT/pytest_gpu_tail.txt
