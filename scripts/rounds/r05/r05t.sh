#!/bin/bash
# look-back-1 window of 64 records where a stream has many workgroups in flight: w1 / w4 on cfg 4
# (4 streams), uniform (4), clipped, 3 components, and ONE cfg-3 frame (a stream alone on the chip)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r05t; mkdir -p $O
cd $REPO
python scripts/exp_ab.py run --what cfg4 w1 w4 w1 w4 w1 w4 2>&1 | grep -v "overhang" > $O/ab_cfg4.txt
cat $O/ab_cfg4.txt
for w in clipped uniform ljpeg3; do
  python scripts/exp_ab.py run --what $w w1 w4 w1 w4 2>&1 | sed "s/^/$w /" | tee -a $O/ab_other.txt
done
for r in 1 2 3; do for v in w1 w4; do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so python bench_ljpeg.py --only cfg3 --frames 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('one_frame $v', j['ms_per_step'], j.get('bit_exact'), j.get('kernels_ms'))" | tee -a $O/ab_one_frame.txt
done; done
for r in 1 2; do for v in w1 w4; do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_$v.so python bench_ljpeg.py --only cfg3 --frames 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('two_frames $v', j['ms_per_step'], j.get('bit_exact'), j.get('kernels_ms'))" | tee -a $O/ab_one_frame.txt
done; done
