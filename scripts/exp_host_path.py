"""PCIe-inclusive rate of the host-pointer entry points (what the patched
reference methods call): pageable host input -> H2D -> kernels -> D2H into the
RawImage buffer.  Never the bench `value`; reported in DESIGN.md."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rawspeed_amd import abi, capi, synth  # noqa: E402
from oracle_lib import HostImage  # noqa: E402
import bench_ljpeg  # noqa: E402


def best(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


ctx = capi.Context(0)
W, H = 8192, 5464
pix = synth.uniform(W * H, 14, seed=3).reshape(H, W)
packed = synth.pack_rows(pix, 14, abi.ORDER_MSB)
d = abi.UnpackDesc(0, 0, W, H, W * 14 // 8, 14, abi.ORDER_MSB)
img = HostImage(W, H, 1)
assert ctx.unpack_u16(d, packed, img.view()) == 0
assert np.array_equal(img.pixels(), pix)
t = best(lambda: ctx.unpack_u16(d, packed, img.view()))
print("rsx_unpack_u16 host call, 8192x5464 14-bit: %.2f ms = %.1f GPix/s (%.1f GB/s over the link)"
      % (t * 1e3, W * H / t / 1e9, (packed.size + W * H * 2) / t / 1e9))

dc, data, src, scan_len, bits = bench_ljpeg.make_cr2_frame(6720, 4480, (3, 2240, 2240), seed=1)
img = HostImage(6720, 4480, 1)
st = ctx.cr2_decode(dc, data, img.view())
assert st[0] == 0 and np.array_equal(img.pixels(), src)
t = best(lambda: ctx.cr2_decode(dc, data, img.view()))
print("rsx_cr2_decode host call, 6720x4480: %.2f ms = %.2f GPix/s" % (t * 1e3, 6720 * 4480 / t / 1e9))
