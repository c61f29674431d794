#!/bin/bash
# Round 6, second GPU session: the reproducer with its own munmap bug fixed (r06a's "memory access faults"
# were the reproducer unmapping a page past its mapping), numpy's allocation pattern added; what the runtime
# does for a 2-D copy into pageable memory (AMD_LOG_LEVEL=4); a long soak of the round-5 failing recipe on
# the library built to take the round-5 2-D copies and to check every download against the device rows.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06b; mkdir -p $O
cd $REPO
R=$REPO/scripts/repro/memcpy2d_pageable
echo "== what the runtime does for one image (rect, off the 16-byte grid)" | tee $O/runtime_trace.txt
AMD_LOG_LEVEL=4 timeout 60 $R rect none malloc 1 0.3 2>&1 | grep -v "amdgpu.ids" > $O/trace_rect_raw.txt
grep -o -i -E "(copyBufferRect[A-Za-z]*|readBufferRect|Unpinned[a-z ]*|pin[a-zA-Z ]*host[a-zA-Z ]*|staging[a-zA-Z ]*|__amd_rocclr_[A-Za-z]+|hsa_amd_memory_[a-z_]+|hipMemcpy2DAsync|SDMA[a-zA-Z ]*|blit[a-zA-Z ]*)" $O/trace_rect_raw.txt | sort | uniq -c | sort -rn | head -40 | tee -a $O/runtime_trace.txt
echo "-- rect16" | tee -a $O/runtime_trace.txt
AMD_LOG_LEVEL=4 timeout 60 $R rect16 none malloc 1 0.3 2>&1 | grep -o -i -E "(copyBufferRect[A-Za-z]*|readBufferRect|__amd_rocclr_[A-Za-z]+|hsa_amd_memory_[a-z_]+|SDMA[a-zA-Z ]*)" | sort | uniq -c | sort -rn | head -20 | tee -a $O/runtime_trace.txt
echo "-- rows1d" | tee -a $O/runtime_trace.txt
AMD_LOG_LEVEL=4 timeout 60 $R rows1d none malloc 1 0.3 2>&1 | grep -o -i -E "(copyBufferRect[A-Za-z]*|readBuffer[A-Za-z]*|__amd_rocclr_[A-Za-z]+|hsa_amd_memory_[a-z_]+|SDMA[a-zA-Z ]*)" | sort | uniq -c | sort -rn | head -20 | tee -a $O/runtime_trace.txt
head -c 300000 $O/trace_rect_raw.txt | tail -c 60000 > $O/trace_rect_excerpt.txt; rm -f $O/trace_rect_raw.txt
echo "== reproducer" | tee $O/repro.txt
for cfg in "rect none mmap 1 30" "rect none numpy 1 40" "rect none numpy 4 40" "rect none mmap 4 30" \
           "rect collapse numpy 1 30" "rect move numpy 1 30" "rect16 none numpy 4 20" \
           "rows1d none numpy 4 20" "pinned none numpy 4 20" "pinned none mmap 4 20"; do
  timeout 120 $R $cfg 2>&1 | grep -v "amdgpu.ids" | tail -6 | cut -c1-400 | tee -a $O/repro.txt
done
echo "== library, round-5 copies + check (diag2d), the recipe of scripts/rounds/r05/r05zz.sh" | tee $O/diag.txt
timeout 100 python scripts/fuzz_more.py big 0 30 2>&1 | tail -1
timeout 100 python scripts/fuzz_more.py big2 0 30 2>&1 | tail -1
for r in 1 2 3 4 5 6 7 8; do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 150 python scripts/fuzz_more.py big3 0 40 2>&1 | grep -v "amdgpu.ids" | grep "FAILED\|RSX_DIAG\|failed:" | cut -c1-500 | tee -a $O/diag.txt
done
grep -E "thp_collapse_alloc |thp_fault_alloc |compact_migrate_scanned|pgmigrate_success|compact_stall" /proc/vmstat | tee $O/vmstat.txt
