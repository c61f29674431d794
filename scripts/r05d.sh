#!/bin/bash
# round 5, GPU call 4: two-table phases; new bench legs
O=gpurun_out/r05d; mkdir -p $O
WHAT=cfg4mt RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_stats.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_cfg4mt.txt
python bench_ljpeg.py --only pentax --no-cpu > $O/pentax.json 2> $O/pentax.err
python bench_ljpeg.py --only samsung_v1 > $O/samsung_v1.json 2> $O/samsung_v1.err
python bench_ljpeg.py --only cfg4 --no-cpu > $O/cfg4.json 2> $O/cfg4.err
grep -v "stream [0-9]" $O/phases_cfg4mt.txt | head -40; tail -30 $O/pentax.json; tail -5 $O/pentax.err; tail -30 $O/samsung_v1.json; tail -3 $O/samsung_v1.err
