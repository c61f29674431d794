#!/bin/bash
# round 5, GPU call 12: scalar-cache invalidation as a kernel of its own; final hand-over polls
O=gpurun_out/r05l; mkdir -p $O
python scripts/exp_ab.py run --what cfg3 r5a base r5a base r5a base > $O/ab_cfg3.txt 2>&1
python scripts/exp_ab.py run --what cfg4 r5a base polls64 r5a base polls64 > $O/ab_cfg4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_ljpeg.py tests/test_gpu_two_tables.py tests/test_gpu_dropin.py tests/test_gpu_raw_files.py -x -q -m gpu > $O/pytest.txt 2>&1
cat $O/ab_cfg3.txt $O/ab_cfg4.txt; tail -3 $O/pytest.txt
