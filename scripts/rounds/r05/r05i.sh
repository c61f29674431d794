#!/bin/bash
# round 5, GPU call 9: restart intervals laid out on the device
O=gpurun_out/r05i; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
python scripts/exp_ab.py run --what cfg4 r5a base r5a base > $O/ab_cfg4.txt 2>&1
cat $O/ab_cfg4.txt
