#!/bin/bash
# round 5, GPU call 6: re-measure after the LDS layout fix; pinned host path
O=gpurun_out/r05f; mkdir -p $O
python scripts/exp_ab.py run --what cfg4 r5a base noresume r5a base noresume > $O/ab_cfg4.txt 2>&1
python scripts/exp_ab.py run --what uniform r5a base noresume r5a base noresume > $O/ab_uniform.txt 2>&1
python scripts/exp_ab.py run --what cfg3 r5a base r5a base > $O/ab_cfg3.txt 2>&1
python bench_ljpeg.py --only host > $O/host.json 2> $O/host.err
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_unpack.py tests/test_gpu_two_tables.py -x -q -m gpu > $O/pytest.txt 2>&1
cat $O/ab_cfg4.txt $O/ab_uniform.txt $O/ab_cfg3.txt; tail -60 $O/host.json; tail -3 $O/host.err; tail -3 $O/pytest.txt
