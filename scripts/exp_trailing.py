"""tests/test_gpu_fast_path.py::test_trailing_bytes_after_the_last_symbol_are_nobodys_business with
the experiment build's statistics (RSX_LIB=<stats build> RSX_DEBUG=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench_ljpeg as B
import cases as C
from rawspeed_amd import abi, capi
gpu = capi.Context(0)
for tail in ("zeros", "pattern"):
    rng = np.random.default_rng([13, len(tail)])
    W, H = 2048, 700
    d, data, px, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1))
    extra = {"zeros": np.zeros(200000, np.uint8),
             "pattern": np.tile(np.array([0x55, 0xAA, 0x3C], np.uint8), 70000)}[tail]
    data = np.concatenate([data, extra])
    j = abi.LJpegJob()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = B.out_pitch(W), W, H, 1, 1
    plan = gpu.ljpeg_plan([j])
    inp = torch.from_numpy(data).cuda()
    out = torch.zeros(B.out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    plan.set_timing(True)
    for run in range(2):
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        tab = plan.kernel_table()
        print("RUN", tail, rc, np.array_equal(B.gpu_frame(out, 0, W, H), px), [k for k, _ in tab[0]] if tab else [], flush=True)
