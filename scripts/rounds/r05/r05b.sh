#!/bin/bash
# round 5, GPU call 2: the head of the single-pass kernel (block index as ticket, arithmetic
# (stream, block), unconditional loads in one batch), LDS-only barriers, nt pixel stores
O=gpurun_out/r05b; mkdir -p $O
python scripts/exp_ab.py run --what cfg3 r4 base tk plainbar nt rot0 rot2 nouni r4 base > $O/ab_cfg3.txt 2>&1
python scripts/exp_ab.py run --what cfg4 r4 base tk plainbar nt rot2 nouni r4 base > $O/ab_cfg4.txt 2>&1
RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_stats.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_cfg3.txt
WHAT=cfg4 RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_stats.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_cfg4.txt
timeout 900 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_two_tables.py tests/test_gpu_baseline_parity.py -x -q -m gpu > $O/pytest.txt 2>&1
cat $O/ab_cfg3.txt $O/ab_cfg4.txt; head -32 $O/phases_cfg3.txt; tail -5 $O/pytest.txt
