#!/usr/bin/env python3
"""profiles/rNN/unpack_pmc.json from the two rocprofv3 --pmc passes
(scripts/collect_profiles.sh): mean FETCH_SIZE / WRITE_SIZE (KB) per launch of the
headline kernel, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys

d = sys.argv[1]


def mean(counter):
    rows = list(csv.DictReader(open("%s/unpack_pmc_%s.csv" % (d, counter))))
    v = [float(r["Counter_Value"]) for r in rows
         if "unpack_kernel<1, 0>" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)


f, nf = mean("FETCH_SIZE")
w, nw = mean("WRITE_SIZE")
alg = 8 * 5464 * (8192 * 14 // 8 + 8192 * 2)
out = {
    "kernel": "unpack_kernel<1, 0> (BitOrder::MSB), 8 frames of 8192x5464 14-bit per launch, "
              "non-temporal loads/stores",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python bench.py "
               "--steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-cfg5 (two separate passes, "
               "scripts/collect_profiles.sh)",
    "launches": [nf, nw],
    "FETCH_SIZE_KB_per_launch": f,
    "WRITE_SIZE_KB_per_launch": w,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read "
                  "(MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE taken as is",
    "read_bytes_per_launch": f * 1024 * 2,
    "write_bytes_per_launch": w * 1024,
    "traffic_bytes_per_launch": f * 1024 * 2 + w * 1024,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (f * 1024 * 2 + w * 1024) / alg,
}
json.dump(out, open("%s/unpack_pmc.json" % d, "w"), indent=1)
print(json.dumps(out, indent=1))
