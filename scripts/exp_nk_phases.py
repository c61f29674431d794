"""Phases of the single-pass kernel on the Nikon leg (experiment build: RSX_LIB = the stats
variant, RSX_DEBUG=1 prints the phase means; RSX_NO_FAST_NK=1 takes the differences route)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench_ljpeg as B
import nikon_cases as N
from rawspeed_amd import abi, capi, synth
ctx = capi.Context(0)
W, H, bits = 6016, 4016, 14
frames = int(os.environ.get("FRAMES", "8"))
unc = os.environ.get("UNCORRECTED", "1") == "1"
src = synth.sensor_image(W, H, 14, seed=8)
meta = N.metadata(70, 0, [2000, 2500, 2500, 3000])
P = N.parse(meta, bits, H)
pu = P["p_up"]
data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]], synth.NIKON_TREE[P["huff_select"]])
data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
d = N.desc(P, bits, unc)
jobs = []
for f in range(frames):
    j = abi.NikonJob()
    j.desc = d
    j.in_offset, j.in_bytes = f * data.size, data.size
    j.img_offset = f * B.out_pitch(W) * H
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = B.out_pitch(W), W, H, 1, 1
    jobs.append(j)
inp = torch.from_numpy(np.tile(data, frames)).cuda()
outb = torch.zeros(frames * B.out_pitch(W) * H, dtype=torch.uint8, device="cuda")
plan = ctx.nikon_plan(jobs)
for _ in range(3):
    plan.run(inp.data_ptr(), outb.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(plan.results()[0])
