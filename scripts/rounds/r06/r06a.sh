#!/bin/bash
# Round 6, first GPU session: (1) the librsx-free reproducer of the undelivered-16-bytes defect in every
# copy mode x stress mode, (2) the round-5 failing recipe (big3 behind other processes' work) on a library
# built to take the round-5 2-D copies and to check every download against the device rows
# (-DRSX_DIAG_DOWNLOAD), (3) the state of the suite and of the bench on this box.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06a; mkdir -p $O
cd $REPO
{
  echo "== environment"
  uname -r
  cat /opt/rocm/.info/version 2>/dev/null
  echo "numa nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)"
  echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>/dev/null)"
  echo "thp enabled: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)"
  echo "thp defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null)"
  echo "khugepaged defrag: $(cat /sys/kernel/mm/transparent_hugepage/khugepaged/defrag 2>/dev/null) scan_sleep_ms: $(cat /sys/kernel/mm/transparent_hugepage/khugepaged/scan_sleep_millisecs 2>/dev/null)"
  echo "compaction proactiveness: $(cat /proc/sys/vm/compaction_proactiveness 2>/dev/null)"
  nproc
  grep -E "MemTotal|AnonHugePages|HugePages_Total" /proc/meminfo
  grep -E "thp_collapse_alloc |compact_migrate_scanned|numa_pages_migrated|pgmigrate_success|numa_hint_faults " /proc/vmstat
} > $O/env.txt 2>&1
cat $O/env.txt
R=$REPO/scripts/repro/memcpy2d_pageable
if [ ! -x $R ]; then hipcc --offload-arch=gfx950 -O2 -o $R $R.hip -lpthread; fi
echo "== reproducer" | tee $O/repro.txt
for cfg in "rect none mmap 1 25" "rect none malloc 1 20" "rect none mmap 4 25" \
           "rect collapse mmap 1 20" "rect move mmap 1 20" "rect fork mmap 1 15" \
           "rect16 collapse mmap 1 15" "rect16 move mmap 1 15" \
           "rows1d collapse mmap 1 15" "rows1d move mmap 1 15" "rows1d none mmap 4 15" \
           "pinned collapse mmap 1 15" "pinned move mmap 1 15" "pinned none mmap 4 15"; do
  timeout 90 $R $cfg 2>&1 | grep -v "amdgpu.ids" | tail -8 | cut -c1-400 | tee -a $O/repro.txt
done
grep -E "thp_collapse_alloc |compact_migrate_scanned|numa_pages_migrated|pgmigrate_success|numa_hint_faults " /proc/vmstat | tee -a $O/env.txt
echo "== library, round-5 copies + check (diag2d)" | tee $O/diag.txt
S="0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32 33 34 35 36 37 38 39"
timeout 100 python scripts/fuzz_more.py big 0 30 2>&1 | tail -1
timeout 100 python scripts/fuzz_more.py big2 0 30 2>&1 | tail -1
for r in 1 2 3; do
  RSX_LIB=$REPO/rawspeed_amd/variants/librsx_diag2d.so timeout 150 python scripts/fuzz_diag.py big3 $S 2>&1 | grep -v "amdgpu.ids\| ok$" | cut -c1-500 | tee -a $O/diag.txt
done
grep -E "thp_collapse_alloc |compact_migrate_scanned|numa_pages_migrated|pgmigrate_success|numa_hint_faults " /proc/vmstat | tee -a $O/env.txt
echo "== suite"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
echo "== bench"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
cp bench_extra.json $O/ 2>/dev/null
