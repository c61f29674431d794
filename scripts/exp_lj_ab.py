"""A/B of librsx variants on the cfg-3 / Nikon legs (interleaved runs)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vdir = os.path.join(ROOT, "rawspeed_amd", "variants")
variants = ["default"] + sorted(f[7:-3] for f in os.listdir(vdir) if f.endswith(".so"))
res = {v: [] for v in variants}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for v in variants:
        env = dict(os.environ)
        if v != "default":
            env["RSX_LIB"] = os.path.join(vdir, "librsx_%s.so" % v)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_ljpeg.py"), "--only", "cfg3"],
                             env=env, capture_output=True, text=True).stdout
        j = json.loads(out[out.index("{"):])
        res[v].append((j["ms_per_step"], j["dominant_kernel"]["avg_ms"], j["bit_exact"]))
for v in variants:
    print(v, res[v])
