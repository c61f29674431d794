#!/usr/bin/env python3
"""Per-kernel HBM bytes of one cfg-3 pipeline run (8 frames of 6720x4480) from the
two rocprofv3 --pmc passes of scripts/pmc_ljpeg_traffic.sh.  FETCH_SIZE / WRITE_SIZE
are in KB; FETCH_SIZE x2 on gfx950 as for the unpack kernel (MI355X_MICROARCH.md,
HBM section) -- the un-stuffed image and the differences are wide streaming reads."""
import collections
import csv
import json
import sys

d = sys.argv[1]


def per_kernel(counter):
    """Steady state only: a plan's FIRST run launches the single-pass kernel at every LDS level
    it might need (one works, the others return), so the first pipeline run of the process is
    left out -- everything up to the second lj_unstuff_kernel launch."""
    rows = [r for r in csv.DictReader(open("%s/ljpeg_pmc_%s.csv" % (d, counter)))
            if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen_unstuff, first_steady = 0, None
    for r in rows:
        if "lj_unstuff_kernel" in r["Kernel_Name"]:
            seen_unstuff += 1
            if seen_unstuff == 2:
                first_steady = int(r["Dispatch_Id"]) - 1  # (its lj_init_results_kernel)
                break
    acc = collections.defaultdict(list)
    for r in rows:
        if first_steady is not None and int(r["Dispatch_Id"]) < first_steady:
            continue
        name = r["Kernel_Name"]
        name = name[name.find("lj_"):].split("(")[0] if "lj_" in name else name[:40]
        acc[name].append(float(r["Counter_Value"]))
    # per pipeline run: mean per launch x launches per run
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


f, w = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
runs = min(n for k, (m, n) in f.items() if k.startswith("lj_unstuff"))
out = {"workload": "bench_ljpeg.py --only cfg3 --frames 8 (8 x 6720x4480, 3 CR2 slices)",
       "runs_profiled": runs, "first_run_left_out": True, "kernels": {}}
tot_r = tot_w = 0.0
for k in sorted(f):
    if not k.startswith("lj_"):
        continue
    rd = f[k][0] * 1024 * 2 * f[k][1] / runs
    wr = w.get(k, (0, 0))[0] * 1024 * w.get(k, (0, 1))[1] / runs
    out["kernels"][k] = {"read_MB_per_run": round(rd / 1e6, 1), "write_MB_per_run": round(wr / 1e6, 1)}
    tot_r += rd
    tot_w += wr
alg = 8 * (31472096 + 6720 * 4480 * 2)
out["total_read_MB"] = round(tot_r / 1e6, 1)
out["total_write_MB"] = round(tot_w / 1e6, 1)
out["algorithmic_MB"] = round(alg / 1e6, 1)
out["traffic_over_algorithmic"] = round((tot_r + tot_w) / alg, 2)
json.dump(out, open("%s/ljpeg_traffic.json" % d, "w"), indent=1)
print(json.dumps(out, indent=1))
