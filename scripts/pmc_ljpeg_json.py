#!/usr/bin/env python3
"""profiles/rNN/ljpeg_pmc/ljpeg_pmc.json from the PMC passes of scripts/pmc_ljpeg.sh (cfg 3,
8 frames): per kernel INSTANTIATION the wave-level instruction counts, and the VALU issue
fraction of the two kernels of the single-pass pipeline,
  valu_issue_frac = VALU wave-instructions x cycles per instruction / (1024 SIMDs x kernel cycles),
with the kernel time from `rocprofv3 --kernel-trace --stats` of the same command
(cfg3_kernel_stats.csv next to this directory) and the two issue costs measured by
scripts/ubench/valu_rates2.hip: 2.4 cycles (add/sub/and/or/xor/lshr/ashr/mov) and 4.3
(everything else) -- given as a [low, high] pair; the loops are ~70 % cheap instructions.

Round 6: counters and times are keyed by the FULL instantiation (`lj_fast_kernel<2, false, 0>`),
not by the template's name: round 5's `lj_\\w+` key let the PROBE instantiation of a plan's first
run (round 5's `<2, false, true>`: two idle launches + one real, 138.6 us "average") overwrite the main
kernel's 397.8 us, and the published issue fraction came out as 1.1-2.0.  PROBE instantiations
(third template argument 1; `true` in round 5's files) are left out altogether, and a fraction above 1 is an error."""
import collections
import csv
import json
import os
import re
import sys


def inst_name(full):
    """'void rsx::(anonymous namespace)::lj_fast_kernel<2, false, 0>(rsx::LjArgs, ...)' ->
    'lj_fast_kernel<2, false, 0>' (None for kernels that are not the LJPEG pipeline's)."""
    m = re.search(r"(lj_\w+)(<[^>]*>)?", full)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def is_probe(name):
    """the single-pass kernel's first-run instantiation: lj_fast_kernel<N, TWO, MODE = 1> (round 5:
    a bool, `true`); MODE 2 -- plans laid out on the device -- is a kernel of its own and stays"""
    m = re.match(r"lj_fast_kernel<\s*\d+\s*,\s*\w+\s*,\s*(\w+)\s*>", name)
    return bool(m and m.group(1) in ("true", "1"))


def base_name(name):
    return name.split("<")[0]


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in sorted(os.listdir(d)):
        if not re.match(r"set\d+\.csv", fn):
            continue
        for r in csv.DictReader(open(os.path.join(d, fn))):
            k = inst_name(r["Kernel_Name"])
            if k and not is_probe(k):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    times = {}
    for cand in (os.path.join(d, "..", "cfg3_kernel_stats.csv"), os.path.join(d, "cfg3_kernel_stats.csv")):
        if os.path.exists(cand):
            for r in csv.DictReader(open(cand)):
                k = inst_name(r["Name"])
                if k and not is_probe(k):
                    times[k] = float(r["TotalDurationNs"]) / int(r["Calls"]) * 1e-9
            break
    return acc, times


def summarise(acc, times):
    out = {"workload": "bench_ljpeg.py --only cfg3 --frames 8 (8 x 6720x4480, 3 CR2 slices)",
           "how": "SQ_INSTS_VALU x [2.4, 4.3] cycles / (1024 SIMDs x kernel time x 2.4 GHz); kernel "
                  "time = rocprofv3 --stats average of the same command, per instantiation, the "
                  "first run's PROBE instantiation left out",
           "kernels": {}, "valu_issue_frac": {}}
    for k, cs in sorted(acc.items()):
        e = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
        e["launches_counted"] = max(len(v) for v in cs.values())
        if k in times:
            e["avg_kernel_us"] = round(times[k] * 1e6, 2)
            if "SQ_INSTS_VALU" in e:
                cyc = 1024 * times[k] * 2.4e9
                e["valu_issue_frac"] = [round(e["SQ_INSTS_VALU"] * 2.4 / cyc, 3),
                                        round(e["SQ_INSTS_VALU"] * 4.3 / cyc, 3)]
                if e["valu_issue_frac"][0] > 1.0:
                    raise SystemExit("pmc_ljpeg_json: %s: VALU issue fraction %r > 1 -- counters and "
                                     "kernel time do not belong together" % (k, e["valu_issue_frac"]))
                if base_name(k) in ("lj_fast_kernel", "lj_unstuff_kernel"):
                    out["valu_issue_frac"][k] = e["valu_issue_frac"]
            if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_ACTIVE_INST_LDS"):
                e["lds_conflict_cycles_per_active_lds_cycle"] = round(
                    e["SQ_LDS_BANK_CONFLICT"] / e["SQ_ACTIVE_INST_LDS"], 3)
        out["kernels"][k] = e
    return out


if __name__ == "__main__":
    d = sys.argv[1]
    out = summarise(*load(d))
    json.dump(out, open(os.path.join(d, "ljpeg_pmc.json"), "w"), indent=1)
    print(json.dumps(out["valu_issue_frac"]))
