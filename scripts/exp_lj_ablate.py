"""GPU experiment: ablations of the LJPEG kernels (RSX_ABLATE) -- timing only,
results are wrong by construction when a phase is skipped."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ab in [int(x) for x in (sys.argv[1:] or ['0','2','10','18','34','26','58'])]:
    env = dict(os.environ, RSX_ABLATE=str(ab))
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", "/tmp/abl%d" % ab,
           "-o", "x", "--", sys.executable, os.path.join(ROOT, "scripts", "exp_lj_run.py")]
    subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    import csv
    rows = list(csv.DictReader(open("/tmp/abl%d/x_kernel_stats.csv" % ab)))
    out = {}
    for r in rows:
        for k in ("lj_unstuff", "lj_sync_kernel<false", "lj_decode_kernel", "lj_predict"):
            if k in r["Name"]:
                out[k] = round(float(r["AverageNs"]) / 1e3, 1)
    print(os.environ.get("RSX_LIB", "default").split("/")[-1], "ablate", ab, out, "total", round(sum(out.values()), 1), flush=True)
