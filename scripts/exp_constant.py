"""Worst case for self-synchronisation: constant frames (periodic bit stream).
Times the cfg-3 shape with a constant image against the normal sensor image."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases as C  # noqa: E402
from rawspeed_amd import abi, capi, synth  # noqa: E402


def run(ctx, src, frames=8, steps=3):
    H, W = src.shape
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [W // 3] * 3)
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.NIKON])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 3, W // 3, W // 3
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
    pad = (-(len(scan) + 2)) % 16 + 16
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
    pitch = (W * 2 + 15) // 16 * 16
    jobs = []
    for f in range(frames):
        j = abi.Cr2Job()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = f * data.size, data.size, f * pitch * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = pitch, W, H, 1, 1
        jobs.append(j)
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    out = torch.zeros(frames * pitch * H, dtype=torch.uint8, device="cuda")
    plan = ctx.cr2_plan(jobs)
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.run(inp.data_ptr(), out.data_ptr())
        rc, st, cons = plan.results()
        ts.append(time.perf_counter() - t0)
        assert rc == 0, st
    got = out[:pitch * H].cpu().numpy().view(np.uint16).reshape(H, pitch // 2)[:, :W]
    plan.close()
    return min(ts) * 1e3, bool(np.array_equal(got, src)), len(scan)


if __name__ == "__main__":
    ctx = capi.Context(0)
    W, H = 6720, 4480
    for name, src in (("sensor", synth.sensor_image(W, H, 14, seed=1)),
                      ("constant", np.full((H, W), 16383, np.uint16)),
                      ("half_constant", np.concatenate(
                          [synth.sensor_image(W, H // 2, 14, seed=1),
                           np.full((H - H // 2, W), 0, np.uint16)]))):
        ms, ok, n = run(ctx, src)
        print("%-14s %8.2f ms per 8 frames (run + results), exact=%s, %d scan bytes"
              % (name, ms, ok, n), flush=True)
