#!/bin/bash
# Round 6: where the Nikon-type instantiation's time goes (phase stamps of the stats build), next to the
# differences route on the same frames; Pentax on the new route.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06n; mkdir -p $O
cd $REPO
L=$REPO/rawspeed_amd/variants/librsx_stats.so
for unc in 1 0; do
echo "== nikon-type pixels, uncorrected=$unc" | tee -a $O/phases.txt
UNCORRECTED=$unc RSX_DEBUG=1 RSX_LIB=$L timeout 200 python scripts/exp_nk_phases.py 2>&1 | grep -E "^\[rsx\]   |single-pass phases|K0 phases" | head -40 | tee -a $O/phases.txt
done
echo "== differences route" | tee -a $O/phases.txt
RSX_NO_FAST_NK=1 UNCORRECTED=1 RSX_DEBUG=1 RSX_LIB=$L timeout 200 python scripts/exp_nk_phases.py 2>&1 | grep -E "^\[rsx\]   |single-pass phases|K0 phases" | head -40 | tee -a $O/phases.txt
timeout 300 python -m pytest tests/test_gpu_nikon.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench_ljpeg.py --only pentax 2>&1 | grep -v "amdgpu.ids" | tee $O/bench_pentax.txt | grep -E "mpix_per_s|ms_per_step|lj_|legacy|bit_exact\""
