#!/bin/bash
# round 5, GPU call 10: DRI on device with the fast scan + K0's funnel loads + scalar-cache invalidation
O=gpurun_out/r05j; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python scripts/exp_ab.py run --what cfg4 r5a base r5a base > $O/ab_cfg4.txt 2>&1
python scripts/exp_ab.py run --what cfg3 r5a base > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg4.txt $O/ab_cfg3.txt
