"""A/B driver for experiment builds of the library.

  python scripts/exp_ab.py build name1="-DFLAG ..." name2=...   (here: hipcc cross-compiles)
  python scripts/exp_ab.py run [--what cfg3] name1 name2 ...     (on the GPU box)

`base` = the shipped rawspeed_amd/librsx.so.  Each variant runs in its own process
(RSX_LIB selects the library) and prints ms_per_step + the per-kernel table.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    if mode == "build":
        from rawspeed_amd import build as b
        for spec in sys.argv[2:]:
            name, flags = spec.split("=", 1)
            # -DRSX_EXPERIMENT also switches the round statistics on (atomics in the
            # synchronisation kernels' loops: +0.07 ms on cfg 3).  "name=prod ..." builds
            # without it, for flags that need no experiment code.
            fl = flags.split()
            prod = bool(fl) and fl[0] == "prod"
            out = b.build_variant(name, fl[1:] if prod else ["-DRSX_EXPERIMENT"] + fl)
            print("built", out)
        return
    args = sys.argv[2:]
    what = "cfg3"
    if args and args[0] == "--what":
        what = args[1]
        args = args[2:]
    for name in args:
        env = dict(os.environ)
        if name != "base":
            env["RSX_LIB"] = os.path.join(ROOT, "rawspeed_amd", "variants", "librsx_%s.so" % name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_ljpeg.py"), "--only", what,
                            "--no-cpu"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            # (the JSON is the last thing printed, pretty-printed from a "{" on a line of its own; log
            # lines in front of it may hold braces of their own)
            at = r.stdout.rfind("\n{\n")
            j = json.loads(r.stdout[at + 1:] if at >= 0 else r.stdout[r.stdout.index("{"):])
            k = j.get("kernels_ms", {})
            print("%-14s %.4f ms exact=%s  %s" % (name, j.get("ms_per_step", 0.0), j.get("bit_exact"),
                                                  " ".join("%s=%.3f" % (a.replace("lj_", "").replace("_kernel", ""), v)
                                                           for a, v in k.items())))
            for sub, v in j.items():  # (cfg 4: its variants -- two tables, small tiles, overhang, DRI)
                if isinstance(v, dict) and "ms_per_step" in v:
                    kk = v.get("kernels_ms") or {}
                    print("%-14s   %-26s %.4f ms exact=%s  %s" % (
                        name, sub, v["ms_per_step"], v.get("bit_exact"),
                        " ".join("%s=%.3f" % (a.replace("lj_", "").replace("_kernel", ""), x)
                                 for a, x in kk.items())))
        except Exception as e:  # noqa: BLE001
            print(name, "FAILED", e, r.stdout[-500:], r.stderr[-1500:])


if __name__ == "__main__":
    main()
