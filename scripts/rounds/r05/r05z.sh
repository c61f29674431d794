#!/bin/bash
# K0's head (stream record in one batch, LUT lengths behind the slot loads): base / prev; look-back
# records as 16-byte pairs: lb16 (= base + -DRSX_LF_LB16)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05z; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_two_tables.py tests/test_gpu_ljpeg.py -q -x 2>&1 | tail -3 | tee $O/pytest_base.txt
RSX_LIB=$REPO/rawspeed_amd/variants/librsx_lb16.so timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_fuzz.py tests/test_gpu_fuzz_r05.py tests/test_gpu_two_tables.py -q -x 2>&1 | tail -3 | tee $O/pytest_lb16.txt
python scripts/exp_ab.py run --what cfg3 base prev lb16 base prev lb16 base prev lb16 > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
python scripts/exp_ab.py run --what cfg4 base prev lb16 base prev lb16 2>&1 | grep -v "overhang\|256x256" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
python scripts/exp_ab.py run --what ljpeg3 base prev lb16 base prev lb16 2>&1 | sed "s/^/ljpeg3 /" | tee $O/ab_other.txt
