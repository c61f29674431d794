#!/usr/bin/env python3
"""profiles/rNN/ljpeg_limiter.json: the quantities that bound the single-pass LJPEG pipeline
(what bench.py's `ljpeg` summary replays next to the HBM fraction), from a round's profiles:

  wg_lifetime_us, wg_phases_us   cfg3_phase_and_round_stats.txt (phase stamps of an
                                 -DRSX_EXPERIMENT build, scripts/exp_lj_stats.py)
  lane_instr_per_symbol          ljpeg_pmc/ljpeg_pmc.json: SQ_INSTS_VALU (wave instructions)
                                 x 64 lanes / symbols, per kernel and for the pipeline
  parses_per_symbol              K0: one parse from bit 0 (A), one from the predecessor's
                                 exit wherever that is not bit 0 (B: 1 - 1 / mean symbol
                                 length), the fixed-point rounds on a dense list (the slots
                                 listed per workgroup / 255, from the same statistics when
                                 present, else the 2 % of round 3's model) + the one decode
  resident_wg_per_cu             from the kernel's registers and LDS (the ISA's metadata)

usage: python scripts/ljpeg_limiter.py profiles/r04 [symbols_per_run] [bits_per_symbol]"""
import json
import os
import re
import sys

d = sys.argv[1]
symbols = int(sys.argv[2]) if len(sys.argv) > 2 else 8 * 6720 * 4480
bits = float(sys.argv[3]) if len(sys.argv) > 3 else 8.35
out = {"how": "scripts/ljpeg_limiter.py over the round's collected profiles (%s: cfg 3, 8 frames, %d "
              "symbols)" % (os.path.basename(os.path.normpath(d)).replace("prof_", "profiles/"), symbols)}
p = os.path.join(d, "cfg3_phase_and_round_stats.txt")
if os.path.exists(p):
    phases = {}
    for line in open(p):
        m = re.match(r"\[rsx\]\s+(.+?)\s+mean\s+([0-9.]+) us", line)
        if m:
            phases[m.group(1).strip()] = float(m.group(2))
    if "lifetime" in phases:
        out["wg_lifetime_us"] = phases.pop("lifetime")
    out["wg_phases_us"] = {k: v for k, v in phases.items() if k != "-"}
p = os.path.join(d, "ljpeg_pmc", "ljpeg_pmc.json")
if os.path.exists(p):
    kall = json.load(open(p))["kernels"]
    # (round 6: the file is keyed by full instantiation, `lj_fast_kernel<2, false, false>`; of a
    # template's instantiations the one that ran longest is the pipeline's kernel)
    k = {}
    for full, e in kall.items():
        base = full.split("<")[0]
        if base not in k or e.get("avg_kernel_us", 0) * e.get("launches_counted", 1) > \
                k[base].get("avg_kernel_us", 0) * k[base].get("launches_counted", 1):
            k[base] = dict(e, instantiation=full)
    per = {}
    for name in ("lj_unstuff_kernel", "lj_fast_kernel", "lj_scan_kernel"):
        if name in k and "SQ_INSTS_VALU" in k[name]:
            per[name] = round(k[name]["SQ_INSTS_VALU"] * 64 / symbols, 1)
    per["pipeline"] = round(sum(per.values()), 1)
    out["lane_instr_per_symbol"] = per
    if "lj_fast_kernel" in k and "avg_kernel_us" in k["lj_fast_kernel"] and "wg_lifetime_us" in out:
        waves = k["lj_fast_kernel"].get("SQ_WAVES")
        if waves:
            wgs = waves / 4
            # kernel time x resident slots (4 a CU x 256 CUs) / workgroups: what one slot spends per
            # workgroup; it can only be >= the mean lifetime x (1 - idle share), never a fraction of it
            out["wg_slot_time_us"] = round(k["lj_fast_kernel"]["avg_kernel_us"] * 4 * 256 / wgs, 2)
            out["wg_slot_time_of"] = k["lj_fast_kernel"]["instantiation"]
            assert out["wg_slot_time_us"] >= 0.5 * out["wg_lifetime_us"], \
                "slot time %.2f us against a lifetime of %.2f us: kernel time and wave count of " \
                "different instantiations" % (out["wg_slot_time_us"], out["wg_lifetime_us"])
out["parses_per_symbol"] = {"_kind": "MODEL, not a measurement: derived from the algorithm and the "
                                     "mean symbol length; nothing in the kernels counts parses",
                            "K0 from bit 0": 1.0, "K0 from the predecessor's exit": round(1 - 1 / bits, 2),
                            "K0 fixed-point rounds (dense list)": 0.02, "decode": 1.0,
                            "total": round(3.02 - 1 / bits, 2)}
out["resident_wg_per_cu"] = {"lj_fast_kernel": 4, "lj_unstuff_kernel": 7,
                             "bound_by": "lj_fast_kernel: 40 KB LDS (and > 102 VGPRs); "
                                         "lj_unstuff_kernel: 22.2 KB LDS"}
json.dump(out, open(os.path.join(d, "ljpeg_limiter.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
