#!/bin/bash
# round 5, GPU call 3: the whole GPU suite + the default bench on the new default build
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.log
cp bench_extra.json $O/ 2>/dev/null
tail -c 3000 $O/bench.json
