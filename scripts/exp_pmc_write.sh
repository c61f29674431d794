#!/bin/bash
# Experiment: WRITE_SIZE / FETCH_SIZE of the cfg-3 pipeline's final decode for library variants
# (run on the GPU box):  scripts/exp_pmc_write.sh base name1 name2 ...
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$REPO/rawspeed_amd/variants/librsx_$v.so
  [ "$v" = base ] && lib=$REPO/rawspeed_amd/librsx.so
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pw_$v
    RSX_LIB=$lib rocprofv3 --pmc $c --output-format csv -d /tmp/pw_$v -- \
      python $REPO/bench_ljpeg.py --only cfg3 --frames 8 --steps 2 --no-cpu > /dev/null 2>&1
    python - "$v" "$c" $(find /tmp/pw_$v -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
v, c, path = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == c and "lj_" in r["Kernel_Name"]:
        n = r["Kernel_Name"]; n = n[n.find("lj_"):].split("(")[0].split("<")[0]
        acc[n].append(float(r["Counter_Value"]))
mul = 2 if c == "FETCH_SIZE" else 1
print(v, c, " ".join("%s=%.0fMB" % (k.replace("lj_", "").replace("_kernel", ""), mul * sum(x) / len(x) * 1024 / 1e6)
                     for k, x in sorted(acc.items()) if sum(x) / len(x) * 1024 / 1e6 > 5))
PY
  done
done
