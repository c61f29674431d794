#!/bin/bash
# Round 6, last: the whole GPU suite and the bench line on the final tree.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06zz; mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.log; tail -c 300 $O/bench_full.json; cp bench_extra.json $O/ 2>/dev/null
