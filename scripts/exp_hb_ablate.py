"""GPU experiment: RSX_ABLATE runs of the Hasselblad leg, K1 (pair) time per launch."""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ab in sys.argv[1:]:
    d = "/tmp/hb_%s" % ab
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                    sys.executable, os.path.join(ROOT, "bench_ljpeg.py"), "--only", "hasselblad", "--steps", "3"],
                   env=dict(os.environ, RSX_ABLATE=ab), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        for k in ("lj_sync_kernel<false", "lj_decode_pair", "lj_unstuff"):
            if k in r["Name"]:
                out[k] = (round(float(r["AverageNs"]) / 1e3, 1), r["Calls"])
    print("ablate", ab, out, flush=True)
