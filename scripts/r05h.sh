#!/bin/bash
# round 5, GPU call 7: three components on the single-pass kernel; host-path changes
O=gpurun_out/r05h; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
python scripts/exp_ab.py run --what ljpeg3 base nofast3 base nofast3 > $O/ab_ljpeg3.txt 2>&1


cat $O/ab_ljpeg3.txt $O/ab_cfg4.txt $O/ab_cfg3.txt
