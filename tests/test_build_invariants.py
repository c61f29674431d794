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
            r"\.amdhsa_kernel (\S*lj_fast_kernelILi(\d)ELb([01])ELb([01])E\S*)(.*?)\.end_amdhsa_kernel",
            text, re.S):
        body = m.group(5)
        found[(int(m.group(2)), bool(int(m.group(3))), bool(int(m.group(4))))] = {
            k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
            for k in ("group_segment_fixed_size", "private_segment_fixed_size",
                      "next_free_vgpr")}
    return found


def test_single_pass_kernel_has_no_static_lds(kernel_descriptors):
    # (components, two alternating tables, the first run's instantiation that looks at the LDS
    # level before it asks for anything else)
    assert set(kernel_descriptors) == {(n, mt, probe) for probe in (False, True) for n, mt in
                                       ((1, False), (2, False), (3, False), (4, False), (2, True),
                                        (4, True))}
    for n, k in kernel_descriptors.items():
        assert k["group_segment_fixed_size"] == 0, (n, k)


def test_single_pass_kernel_fits_four_workgroups_per_cu(kernel_descriptors):
    for n, k in kernel_descriptors.items():
        assert k["next_free_vgpr"] <= 128, (n, k)
        assert k["private_segment_fixed_size"] == 0, (n, k)


def test_unstuff_kernel_keeps_seven_workgroups_per_cu():
    """K0 (both instantiations: plans without / with two-table streams): seven workgroups of
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
            r"\.amdhsa_kernel (\S*lj_unstuff_kernelILb([01])E\S*)(.*?)\.end_amdhsa_kernel",
            text, re.S):
        body = m.group(3)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        seen.add(bool(int(m.group(2))))
        assert get("next_free_vgpr") <= 72, (m.group(1), get("next_free_vgpr"))
        assert get("private_segment_fixed_size") == 0
        assert get("group_segment_fixed_size") == 0
    assert seen == {False, True}
