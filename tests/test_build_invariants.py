"""Properties of the COMPILED single-pass LJPEG kernel that its source relies on and no
run-time test would report kindly (no GPU needed: hipcc cross-compiles gfx950):

* no static LDS -- the kernel addresses its 10-bit LUT absolutely ((window >> 19) & 0x1FF8
  is an LDS address), so its dynamic LDS has to start at LDS address 0; an intrinsic that
  brings static LDS along (__syncthreads_or: 256 bytes) shifts everything and the kernel
  decodes garbage until its look-backs time out;
* no scratch for the 1- and 2-component instantiations (BASELINE configs 3-5): the 64
  running sums live in registers at exactly four workgroups per CU.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    pytest.skip("hipcc not found")


@pytest.fixture(scope="module")
def kernel_descriptors():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fast.s")
        subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                        "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "rawspeed_amd", "csrc"),
                        os.path.join(ROOT, "rawspeed_amd", "csrc", "rsx_ljpeg_fast.hip"),
                        "-o", out], check=True, capture_output=True, timeout=300)
        text = open(out).read()
    found = {}
    for m in re.finditer(
            r"\.amdhsa_kernel (\S*lj_fast_kernelILi(\d)ELi(\d)ELi(\d)ELb([01])ELb([01])E\S*)(.*?)\.end_amdhsa_kernel",
            text, re.S):
        body = m.group(7)
        found[(int(m.group(2)), int(m.group(3)), int(m.group(4)),
               "differences" if int(m.group(5)) else ("nikon-type" if int(m.group(6)) else "pixels"))] = {
            k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
            for k in ("group_segment_fixed_size", "private_segment_fixed_size",
                      "next_free_vgpr")}
    return found


def test_single_pass_kernel_has_no_static_lds(kernel_descriptors):
    # (components; tables: 0 one, 1 two alternating, 2 one per phase of the MCU; mode: 0 steady,
    # 1 the first run's instantiation that looks at the LDS level before it asks for anything
    # else, 2 the same with the scalar-cache refresh of device-resident layouts; differences
    # instead of pixels -- the sRaw / Sony / split-Nikon route, one component, one table)
    pixels = {(n, tm, mode, "pixels") for mode in (0, 1, 2) for n, tm in
              ((1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (4, 1), (2, 2), (3, 2), (4, 2))}
    differences = {(1, 0, mode, "differences") for mode in (0, 1, 2)}
    # (a Nikon-type stream's pixels: two column parities, one table; the vertical sums by row parity)
    differences |= {(2, 0, mode, "nikon-type") for mode in (0, 1, 2)}
    assert set(kernel_descriptors) == pixels | differences
    for n, k in kernel_descriptors.items():
        assert k["group_segment_fixed_size"] == 0, (n, k)


def test_single_pass_kernel_fits_four_workgroups_per_cu(kernel_descriptors):
    for n, k in kernel_descriptors.items():
        assert k["next_free_vgpr"] <= 128, (n, k)
        assert k["private_segment_fixed_size"] == 0, (n, k)


def test_unstuff_kernel_keeps_seven_workgroups_per_cu():
    """K0 (its instantiations: plans with one-table streams only / with two alternating tables /
    with a table per phase, each without and with the scalar-cache refresh): seven workgroups of
    four wavefronts a CU are seven wavefronts a SIMD -- at most 72 vector registers each, no
    scratch -- and 18 LDS granules of 1280 bytes (a static_assert in the source checks the
    two-table layout's size)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k0.s")
        subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                        "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "rawspeed_amd", "csrc"),
                        os.path.join(ROOT, "rawspeed_amd", "csrc", "rsx_ljpeg.hip"),
                        "-o", out], check=True, capture_output=True, timeout=600)
        text = open(out).read()
    seen = set()
    for m in re.finditer(
            r"\.amdhsa_kernel (\S*lj_unstuff_kernelILi(\d)ELb([01])E\S*)(.*?)\.end_amdhsa_kernel",
            text, re.S):
        body = m.group(4)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        seen.add((int(m.group(2)), bool(int(m.group(3)))))
        assert get("next_free_vgpr") <= 72, (m.group(1), get("next_free_vgpr"))
        assert get("private_segment_fixed_size") == 0
        assert get("group_segment_fixed_size") == 0
    assert seen == {(km, inv) for km in (0, 1, 2) for inv in (False, True)}


def test_no_hipmemset_is_trusted_to_have_run():
    """hipMemset of device memory returns before it has run on this runtime, and the library's streams
    are hipStreamNonBlocking -- not ordered behind the null stream the memset runs on
    (scripts/repro/memset_async.hip; the round-6 defect of plan creation).  Every group of
    hipMemset calls in the library's sources is followed by a wait for the null stream before anything
    else can be queued; hipMemsetAsync on the stream of the kernels that read the bytes needs none."""
    src = os.path.join(ROOT, "rawspeed_amd", "csrc")
    seen = 0
    for name in sorted(os.listdir(src)):
        if not name.endswith((".hip", ".cpp", ".h")):
            continue
        lines = open(os.path.join(src, name)).read().split("\n")
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            if re.search(r"\bhipMemset\(", code) is None:
                continue
            seen += 1
            tail = "\n".join(l.split("//")[0] for l in lines[i + 1:i + 30])
            assert re.search(r"\bhipMemset\(|hipStreamSynchronize\(nullptr\)|hipDeviceSynchronize\(\)", tail), \
                "%s:%d: hipMemset without a wait for the null stream behind it" % (name, i + 1)
    assert seen >= 6
