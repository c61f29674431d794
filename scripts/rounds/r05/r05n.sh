#!/bin/bash
# round 5, final GPU pass: the whole suite, the default bench, LJPEG traffic counters again
O=gpurun_out/r05n; mkdir -p $O
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.log
cp bench_extra.json $O/
bash scripts/pmc_ljpeg_traffic.sh > /dev/null 2>&1
mkdir -p $O/ljpeg_traffic; cp gpurun_out/pmc_lj_traffic/* $O/ljpeg_traffic/
python scripts/exp_ab.py run --what cfg3 r5a base r5a base > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt; python -c "
import json; d=json.load(open('$O/ljpeg_traffic/ljpeg_traffic.json')); print(d['traffic_over_algorithmic'], {k:v for k,v in d['kernels'].items()})"
tail -c 600 $O/bench_full.json
