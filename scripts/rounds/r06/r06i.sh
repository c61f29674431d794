#!/bin/bash
# Round 6: the LUT gather micro-benchmark; K1f's LDS bank conflicts split by phase (ablation builds under
# rocprofv3 --pmc: whole kernel, without the decode loop, without staging + copy-out, without both);
# table-per-phase streams with twelve rounds.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06i; mkdir -p $O
cd $REPO
./scripts/ubench/lut_gather 2>&1 | grep -v amdgpu.ids | tee $O/ubench_lut_gather.txt
cd /tmp && export TMPDIR=/tmp
for v in abl0 abl16 abl3 abl19; do
  rm -rf /tmp/pc_$v
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pc_$v -- \
    python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
  f=$(find /tmp/pc_$v -name "*counter_collection.csv" | head -1)
  echo "== $v" | tee -a $O/lds_conflicts_by_phase.txt
  python $REPO/scripts/pmc_summary.py $f lj_fast_kernel | grep -A5 "lj_fast_kernel<2, 0, 0>" | tee -a $O/lds_conflicts_by_phase.txt
done
cd $REPO
timeout 600 python -m pytest tests/test_gpu_per_component_tables.py tests/test_gpu_two_tables.py tests/test_gpu_fast_path.py -q 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python bench_ljpeg.py --only ljpegpt 2>/dev/null | grep -E "ms_per_step|\"lj_|bit_exact\"" | tee $O/bench_pt.txt
