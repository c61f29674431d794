"""rsx_cr2_decode on a cfg-3 frame and rsx_dng_decompress_ljpeg on the cfg-4 tiles, host pointers: ms a call,
how many calls ran in chunks (GPU box).  RSX_HOST_NO_OVERLAP=1: the plain way."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench_ljpeg as B
from rawspeed_amd import capi
from oracle_lib import HostImage
ctx = capi.Context(0)
W, H = 6720, 4480
m = B.make_cr2_frame(W, H, (3, 2240, 2240), seed=1); d, data, px = m[0], m[1], m[2]
img = HostImage(W, H)
for _ in range(3):
    st = ctx.cr2_decode(d, data, img.view())
assert st[0] == 0 and np.array_equal(img.pixels(), px)
c0 = ctx.chunked_calls()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); ctx.cr2_decode(d, data, img.view()); ts.append(time.perf_counter() - t0)
print("cfg3 rsx_cr2_decode: median %.3f ms  min %.3f ms  chunked %d of 30  (input %.1f MB, image %.1f MB)"
      % (1e3 * sorted(ts)[15], 1e3 * min(ts), ctx.chunked_calls() - c0, data.size / 1e6, W * H * 2 / 1e6))
assert np.array_equal(img.pixels(), px)
W4, H4 = 8192, 5464
src, jobs, datas, blobs, lens = B._dng_tiles(W4, H4, 4096, 2732, 2, 0)
img4 = HostImage(W4, H4)
descs = [j.desc for j in jobs]
for _ in range(3):
    rc = ctx.dng_decompress_ljpeg(descs, datas, img4.view())
assert rc[0] == 0 and np.array_equal(img4.pixels(), src)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); ctx.dng_decompress_ljpeg(descs, datas, img4.view()); ts.append(time.perf_counter() - t0)
print("cfg4 rsx_dng_decompress_ljpeg: median %.3f ms  min %.3f ms" % (1e3 * sorted(ts)[15], 1e3 * min(ts)))

# one LJPEG frame as ONE scan (a cfg-5 frame: SOF 4096 x 5464, 2 components): full-width rows come back
import cases as C
from rawspeed_amd import abi, synth
W5, H5 = 8192, 5464
src5 = synth.sensor_image(W5, H5, 14, seed=5)
rows = C.ljpeg_stream_rows(src5, 2, 1, W5 // 2, H5, np.random.default_rng(1), 14)
scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()], 0, False)
d5 = abi.LJpegDesc()
d5.tile_x, d5.tile_y, d5.tile_w, d5.tile_h = 0, 0, W5, H5
d5.mcu_w, d5.mcu_h, d5.frame_w, d5.frame_h = 2, 1, W5 // 2, H5
d5.n_comp, d5.rows_per_restart_interval = 2, H5
abi.fill_recipe(d5, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
data5 = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(64, np.uint8)])
img5 = HostImage(W5, H5)
for _ in range(3):
    st = ctx.ljpeg_decode(d5, data5, img5.view())
assert st[0] == 0 and np.array_equal(img5.pixels(), src5)
c0 = ctx.chunked_calls()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); ctx.ljpeg_decode(d5, data5, img5.view()); ts.append(time.perf_counter() - t0)
print("one 8192x5464 LJPEG scan, rsx_ljpeg_decode: median %.3f ms  min %.3f ms  chunked %d of 30  (input %.1f MB, image %.1f MB)"
      % (1e3 * sorted(ts)[15], 1e3 * min(ts), ctx.chunked_calls() - c0, data5.size / 1e6, W5 * H5 * 2 / 1e6))
