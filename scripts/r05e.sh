#!/bin/bash
# round 5, GPU call 5: long codes from LDS + stopped lanes resume in place
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_two_tables.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_parity.py tests/test_gpu_ljpeg.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
python scripts/exp_ab.py run --what cfg4 r5a base noresume r5a base > $O/ab_cfg4.txt 2>&1
python scripts/exp_ab.py run --what uniform r5a base noresume > $O/ab_uniform.txt 2>&1
python scripts/exp_ab.py run --what cfg3 r5a base r5a base > $O/ab_cfg3.txt 2>&1
python scripts/exp_ab.py run --what clipped r5a base > $O/ab_clipped.txt 2>&1
WHAT=cfg4mt RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_stats.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_cfg4mt.txt
cat $O/ab_cfg4.txt $O/ab_uniform.txt $O/ab_cfg3.txt $O/ab_clipped.txt; grep -v "stream [0-9]" $O/phases_cfg4mt.txt | head -30
