#!/bin/bash
# round 5, GPU call 1: residency slope, no-ticket head, copy-out / look-back ablations
O=gpurun_out/r05a; mkdir -p $O
python scripts/exp_ab.py run --what cfg3 base lds3 lds2 notk base notk stats nost nocopy nolb1 statsnotk > $O/ab_cfg3.txt 2>&1
python scripts/exp_ab.py run --what cfg4 base lds3 lds2 notk base notk > $O/ab_cfg4.txt 2>&1
RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_stats.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_base.txt
RSX_DEBUG=1 RSX_LIB=rawspeed_amd/variants/librsx_statsnotk.so python scripts/exp_lj_stats.py 2>&1 | grep "^\[rsx\]" | cut -c1-300 > $O/phases_notk.txt
cat $O/ab_cfg3.txt $O/ab_cfg4.txt; head -20 $O/phases_base.txt; head -20 $O/phases_notk.txt
