"""In-tree builds of the native libraries (no JIT cache, no site-packages).

  rawspeed_amd/librsx.so        HIP kernels + C-ABI, hipcc --offload-arch=gfx950
  rawspeed_amd/librsx_synth.so  host-side stream writers (gcc)

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container
as well as on the MI355X box.  A library is rebuilt only when a source is newer.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")

LIB_CORE = os.path.join(PKG, "librsx.so")
LIB_SYNTH = os.path.join(PKG, "librsx_synth.so")

CORE_SOURCES = ["rsx_api.hip", "rsx_unpack.hip", "rsx_ljpeg.hip", "rsx_ljpeg_direct.hip",
                "rsx_ljpeg_fast.hip", "rsx_ljpeg_recon.hip", "rsx_sraw.hip", "rsx_samsung_v2.hip",
                "rsx_host.cpp"]
CORE_HEADERS = ["rsx_internal.h", "rsx_device.h", "rsx_ljpeg.h", "rsx_ljpeg_dev.h",
                "rsx_ljpeg_bits.h", "rsx_samsung_v2.h"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_synth(force=False):
    src = os.path.join(CSRC, "synth", "rsx_synth.c")
    if force or _stale(LIB_SYNTH, [src]):
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra",
              "-o", LIB_SYNTH, src])
    return LIB_SYNTH


def _compile_objects(objdir, extra_flags, force=False):
    """One object per source under `objdir`, rebuilt when the source or any header is newer;
    the stale ones in parallel (the sources are independent translation units: one hipcc run
    over all of them took a minute whatever had changed)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in CORE_HEADERS] + [os.path.join(INCLUDE, "rsx.h")]
    jobs, objs = [], []
    for name in CORE_SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                         "-Wall", "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC,
                         *extra_flags, "-o", obj, src])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    return objs, bool(jobs)


def _link(out, objs):
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])


def build_core(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in CORE_SOURCES]
    srcs = [s for s in srcs if os.path.exists(s)]
    deps = srcs + [os.path.join(CSRC, h) for h in CORE_HEADERS] + \
        [os.path.join(INCLUDE, "rsx.h")]
    if force or _stale(LIB_CORE, deps):
        objs, _ = _compile_objects(os.path.join(PKG, "_build", "core"), extra_flags, force)
        _link(LIB_CORE, objs)
    return LIB_CORE


def build_variant(name, extra_flags):
    """A/B builds: rawspeed_amd/variants/librsx_<name>.so (git-ignored; load with
    RSX_LIB=<path>)."""
    d = os.path.join(PKG, "variants")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "librsx_%s.so" % name)
    import hashlib
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]  # (other flags, other objects)
    objs, _ = _compile_objects(os.path.join(PKG, "_build", "variant_%s_%s" % (name, tag)), extra_flags)
    _link(out, objs)
    return out


def build_all(force=False):
    return build_core(force), build_synth(force)


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv))
