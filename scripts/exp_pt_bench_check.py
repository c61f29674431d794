"""The 4-component A B C D bench leg, tile by tile: which tiles / consumed counts differ (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from rawspeed_amd import abi, capi
import bench_ljpeg as B
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cpp = 3 if n == 3 else 1
W, H, tw, th = 8192, 5464, 4096, 2732
pitch = (W * cpp * 2 + 15) // 16 * 16
rng = np.random.default_rng(33)
trng = np.random.default_rng(3303)
tables = (cases.NIKON, cases.ALT) + tuple(cases.random_huffman_table(trng, n_cat=15, skew=1.5) for _ in range(n - 2))
jobs, parts, tiles, off = [], [], [], 0
for ty in range(2):
    for tx in range(2):
        d, data, tile_px, scan_len = cases.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=cpp, tile=(tx * tw, ty * th, tw, th),
                                                           mcu=(n, 1), tables=tables, table_index=list(range(n)))
        pad = (-data.size) % 16
        j = abi.LJpegJob(); j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = pitch, W, H, cpp, int(cpp == 1)
        jobs.append(j); parts.append(np.concatenate([data, np.zeros(pad, np.uint8)])); tiles.append((tx, ty, tile_px, scan_len))
        off += data.size + pad
        print("tile", tx, ty, "px range", tile_px.min(), tile_px.max(), "clipped fraction", float((tile_px == tile_px.max()).mean()))
inp = torch.from_numpy(np.concatenate(parts)).cuda()
out = torch.zeros(pitch * H, dtype=torch.uint8, device="cuda")
plan = ctx.ljpeg_plan(jobs)
s = torch.cuda.current_stream().cuda_stream
for run in range(4):
    out.zero_()
    plan.set_timing(True)
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    rc, st, cons = plan.results()
    tab = plan.kernel_table(); plan.kernel_time()
    got = out.cpu().numpy().view(np.uint16).reshape(H, pitch // 2)[:, :W * cpp]
    print("run", run, "rc", rc, "st", list(st), "consumed ok", [c == t[3] for c, t in zip(cons, tiles)],
          "tiles ok", [bool(np.array_equal(got[ty * th:(ty + 1) * th, tx * tw * cpp:(tx + 1) * tw * cpp], px)) for tx, ty, px, _ in tiles],
          "kernels", [n_ for n_, _ in tab[0]][:8] if tab else None)
