"""Summarise a rocprofv3 counter_collection.csv: per kernel, mean of every counter."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    m = re.search(r"(lj_\w+(<[^>]*>)?|unpack_kernel(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
