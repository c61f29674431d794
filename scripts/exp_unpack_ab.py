"""GPU experiment: A/B builds of the unpack kernel (RSX_LIB) on the bench workload;
interleaved rounds in separate processes, kernel time from the library's hipEvents."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = ["default"] + sorted(f[7:-3] for f in os.listdir(os.path.join(ROOT, "rawspeed_amd", "variants")) if f.endswith(".so"))
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        env = dict(os.environ)
        if v != "default":
            env["RSX_LIB"] = os.path.join(ROOT, "rawspeed_amd", "variants", "librsx_%s.so" % v)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline", "--no-cfg5", "--steps", "30"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
        if not r.stdout.strip():
            print(v, 'FAILED', r.stderr[-800:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        res[v].append((d["roofline"]["avg_kernel_ms"], d["roofline"]["achieved"], d["bit_exact"]))
for v in variants:
    print(v, [x[1] for x in res[v]], "exact", all(x[2] for x in res[v]))
