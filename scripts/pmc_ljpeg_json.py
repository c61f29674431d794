#!/usr/bin/env python3
"""profiles/rNN/ljpeg_pmc/ljpeg_pmc.json from the PMC passes of scripts/pmc_ljpeg.sh (cfg 3,
8 frames): per kernel the wave-level instruction counts, and the VALU issue fraction of
the two kernels of the single-pass pipeline,
  valu_issue_frac = VALU wave-instructions x cycles per instruction / (1024 SIMDs x kernel cycles),
with the kernel time from `rocprofv3 --kernel-trace --stats` of the same command
(cfg3_kernel_stats.csv next to this directory) and the two issue costs measured by
scripts/ubench/valu_rates2.hip: 2.4 cycles (add/sub/and/or/xor/lshr/ashr/mov) and 4.3
(everything else) -- given as a [low, high] pair; the loops are ~70 % cheap instructions."""
import collections
import csv
import json
import os
import re
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(os.listdir(d)):
    if not re.match(r"set\d+\.csv", fn):
        continue
    for r in csv.DictReader(open(os.path.join(d, fn))):
        m = re.search(r"(lj_\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
times = {}
for cand in (os.path.join(d, "..", "cfg3_kernel_stats.csv"), os.path.join(d, "cfg3_kernel_stats.csv")):
    if os.path.exists(cand):
        rows = [(re.search(r"(lj_\w+)", r["Name"]), r) for r in csv.DictReader(open(cand))]
        runs = max([int(r["Calls"]) for m, r in rows if m and m.group(1) == "lj_unstuff_kernel"] + [0])
        for m, r in rows:
            if not m:
                continue
            calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
            # (the single-pass kernel is launched at every LDS level in a plan's first run:
            # the launches whose workgroups leave at once are not part of the average)
            if runs and calls > runs:
                total -= (calls - runs) * float(r["MinNs"])
                calls = runs
            times[m.group(1)] = total / calls * 1e-9
        break
out = {"workload": "bench_ljpeg.py --only cfg3 --frames 8 (8 x 6720x4480, 3 CR2 slices)",
       "how": "SQ_INSTS_VALU x [2.4, 4.3] cycles / (1024 SIMDs x kernel time x 2.4 GHz); kernel "
              "time = rocprofv3 --stats average of the same command",
       "kernels": {}, "valu_issue_frac": {}}
for k, cs in sorted(acc.items()):
    # (launches whose workgroups leave at once -- the other LDS levels in a plan's first run --
    # are not part of the averages: as many launches as K0 has, the largest ones)
    def mean(c, v):
        n = len(acc.get("lj_unstuff_kernel", {}).get(c, [])) or len(v)
        top = sorted(v, reverse=True)[:min(n, len(v))]
        return round(sum(top) / len(top), 1)
    e = {c: mean(c, v) for c, v in sorted(cs.items())}
    if k in times:
        e["avg_kernel_us"] = round(times[k] * 1e6, 2)
        if "SQ_INSTS_VALU" in e:
            cyc = 1024 * times[k] * 2.4e9
            e["valu_issue_frac"] = [round(e["SQ_INSTS_VALU"] * 2.4 / cyc, 3),
                                    round(e["SQ_INSTS_VALU"] * 4.3 / cyc, 3)]
            if k in ("lj_fast_kernel", "lj_unstuff_kernel"):
                out["valu_issue_frac"][k] = e["valu_issue_frac"]
    out["kernels"][k] = e
json.dump(out, open(os.path.join(d, "ljpeg_pmc.json"), "w"), indent=1)
print(json.dumps(out["valu_issue_frac"]))
