"""Why a two-table stream left the single-pass kernel (experiment build, RSX_DEBUG=1).
CASE=ljpeg (default) | cr2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench_ljpeg as B
import cases as C
from rawspeed_amd import abi, capi
ctx = capi.Context(0)
case = os.environ.get("CASE", "ljpeg")
if case == "cr2":
    n_comp, slices = 2, (3, 1344, 1408)
    rng = np.random.default_rng(31 + n_comp)
    W = (slices[0] - 1) * slices[1] + slices[2]
    H = 900
    d, data, tile_px, _ = C.make_cr2_case(rng, W, H, n_comp, slices, tables=(C.NIKON, C.ALT),
                                          table_index=[0, 1])
    j = abi.Cr2Job()
else:
    rng = np.random.default_rng(5)
    W, H = 2048, 512
    d, data, tile_px, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H),
                                            mcu=(2, 1), tables=(C.ALT, C.NIKON), table_index=[1, 0])
    j = abi.LJpegJob()
print("saturated fraction", float((tile_px == tile_px.max()).mean()), tile_px.max())
j.desc = d
j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
op = B.out_pitch(W)
j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
plan = ctx.cr2_plan([j]) if case == "cr2" else ctx.ljpeg_plan([j])
inp = torch.from_numpy(np.ascontiguousarray(data)).cuda()
out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
print(plan.results())
px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
print("exact", np.array_equal(px, tile_px))
