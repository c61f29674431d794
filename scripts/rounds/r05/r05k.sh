#!/bin/bash
# round 5, GPU call 11: where do the 2-4 % on cfg 3 since commit 22aeab1 come from?
O=gpurun_out/r05k; mkdir -p $O
python scripts/exp_ab.py run --what cfg3 r5a base nofresh nofunnel r5a base nofresh nofunnel r5a base > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
