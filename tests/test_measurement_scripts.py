"""The scripts that turn rocprofv3 output into the figures bench.py replays (scripts/pmc_ljpeg_json.py,
scripts/ljpeg_limiter.py) and bench.py's own guard.  Round 5 published a VALU issue fraction of 1.1-2.0:
the summary keyed kernels by the template's NAME, and the first-run PROBE instantiation's time (two idle
launches + one real) landed under the main instantiation's counters."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    return spec, m


MAIN = "void rsx::(anonymous namespace)::lj_fast_kernel<2, false, 0>(rsx::LjArgs, unsigned int, unsigned int)"
PROBE = "void rsx::(anonymous namespace)::lj_fast_kernel<2, false, 1>(rsx::LjArgs, unsigned int, unsigned int)"
PROBE_R5 = "void rsx::(anonymous namespace)::lj_fast_kernel<2, false, true>(rsx::LjArgs, unsigned int, unsigned int)"
K0 = "void rsx::(anonymous namespace)::lj_unstuff_kernel<false, false>(rsx::LjArgs)"


def _write_profile(d):
    os.makedirs(os.path.join(d, "ljpeg_pmc"))
    with open(os.path.join(d, "cfg3_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerow([MAIN, 15, 5967580, 397838.7, 29.96, 389564, 405164, 4595.2])
        w.writerow([K0, 16, 4318243, 269890.2, 21.68, 263602, 278203, 3999.9])
        w.writerow([PROBE, 3, 415685, 138561.7, 2.09, 5480, 403565, 229500.4])       # listed AFTER the main one
        w.writerow([PROBE_R5, 3, 415685, 138561.7, 2.09, 5480, 403565, 229500.4])
    with open(os.path.join(d, "ljpeg_pmc", "set1.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for _ in range(2):
            w.writerow([MAIN, "SQ_INSTS_VALU", 156715933.9])
            w.writerow([MAIN, "SQ_WAVES", 61728.0])
            w.writerow([K0, "SQ_INSTS_VALU", 99800000.0])
        w.writerow([PROBE, "SQ_INSTS_VALU", 52000000.0])
        w.writerow([PROBE_R5, "SQ_INSTS_VALU", 52000000.0])


def test_pmc_summary_keys_by_instantiation_and_leaves_probe_out(tmp_path):
    d = str(tmp_path / "r99")
    _write_profile(d)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_ljpeg_json.py"),
                        os.path.join(d, "ljpeg_pmc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.load(open(os.path.join(d, "ljpeg_pmc", "ljpeg_pmc.json")))
    ks = out["kernels"]
    assert "lj_fast_kernel<2, false, 0>" in ks and "lj_unstuff_kernel<false, false>" in ks
    assert not any(k.endswith("1>") or k.endswith("true>") for k in ks), list(ks)
    e = ks["lj_fast_kernel<2, false, 0>"]
    assert abs(e["avg_kernel_us"] - 397.84) < 0.01          # NOT the probe's 138.56
    lo, hi = e["valu_issue_frac"]
    assert 0.38 < lo < 0.39 and 0.68 < hi < 0.70            # the review's recomputation: [0.385, 0.689]
    for v in out["valu_issue_frac"].values():
        assert all(0 <= x <= 1 for x in v)


def test_pmc_summary_refuses_a_fraction_above_one(tmp_path):
    d = str(tmp_path / "r98")
    _write_profile(d)
    # the main instantiation's time replaced by the probe's: what round 5's key did
    p = os.path.join(d, "cfg3_kernel_stats.csv")
    s = open(p).read().replace('"15","5967580"', '"15","2078430"')
    open(p, "w").write(s)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_ljpeg_json.py"),
                        os.path.join(d, "ljpeg_pmc")], capture_output=True, text=True)
    assert r.returncode != 0 and "> 1" in (r.stderr + r.stdout)


def test_limiter_labels_parse_counts_as_a_model(tmp_path):
    d = str(tmp_path / "r97")
    _write_profile(d)
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_ljpeg_json.py"),
                    os.path.join(d, "ljpeg_pmc")], check=True, capture_output=True)
    with open(os.path.join(d, "cfg3_phase_and_round_stats.txt"), "w") as f:
        f.write("[rsx]   decode             mean    7.47 us  p50   7.46\n[rsx]   lifetime       mean   23.90 us\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ljpeg_limiter.py"), d],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.load(open(os.path.join(d, "ljpeg_limiter.json")))
    assert "MODEL" in out["parses_per_symbol"]["_kind"]
    # kernel time x 1024 resident slots / workgroups: 397.84 us x 1024 / 15432 = 26.4 us (round 5 said 9.19)
    assert abs(out["wg_slot_time_us"] - 26.4) < 0.1
    assert out["lane_instr_per_symbol"]["lj_fast_kernel"] > 40


def test_bench_does_not_replay_a_fraction_above_one(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles" / "r99" / "ljpeg_pmc"
    prof.mkdir(parents=True)
    (prof / "ljpeg_pmc.json").write_text(json.dumps(
        {"how": "x", "valu_issue_frac": {"lj_fast_kernel": [1.105, 1.979], "lj_unstuff_kernel": [0.36, 0.65]}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rep = bench.replayed_ljpeg_counters()
    assert "valu_issue_frac" not in rep
    assert "r99" in rep["valu_issue_frac_rejected"]
