#!/bin/bash
# Round 6, last GPU minutes: the round-5 recipe on the diagnostic library (the round-5 2-D copies + every
# download compared with the device rows, with the checks A0 / A / B / C / D on an event) once more, eight
# processes in sequence and then six at a time, for as long as the budget allows.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06diag; mkdir -p $O
cd $REPO
T0=$(date +%s)
n=0
while [ $(( $(date +%s) - T0 )) -lt 1080 ]; do
  n=$((n+1))
  if [ $n -le 8 ]; then
    RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 150 python scripts/fuzz_more.py big3 0 40 2>&1 | grep -v "amdgpu.ids" | grep "FAILED\|RSX_DIAG\|failed:" | cut -c1-700 >> $O/diag.txt
  else
    for k in 1 2 3 4 5 6; do
      ( RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 300 python scripts/fuzz_more.py big3 0 40 2>&1 | grep -v "amdgpu.ids" | grep "FAILED\|RSX_DIAG\|failed:" | cut -c1-700 > $O/par_$k.txt ) &
    done
    wait
    cat $O/par_*.txt >> $O/diag.txt
  fi
done
echo "processes: $(grep -c 'failed:' $O/diag.txt)"; grep -c "RSX_DIAG_DOWNLOAD: rect" $O/diag.txt; grep "RSX_DIAG_DOWNLOAD" $O/diag.txt | grep -v "row .*host != device" | head -40
