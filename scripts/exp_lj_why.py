"""Why the single-pass LJPEG kernel hands streams over (experiment build, RSX_DEBUG=1):
   RSX_LIB=rawspeed_amd/variants/librsx_exp.so RSX_DEBUG=1 python scripts/exp_lj_why.py [W H]
Prints the plan's per-stream counters (stderr) for a frame with blown highlights and a
black border, two runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench_ljpeg as B
import cases as C
from rawspeed_amd import abi, capi, synth
ctx = capi.Context(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4480
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ns = 2
made = []
for f in range(2):
    src = B.clipped_image(W, H, int(os.environ.get("SEED", "70")) + f)
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, C.cr2_slices(ns, W // ns, W // ns))
    scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = ns, W // ns, W // ns
    abi.fill_recipe(d, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
    pad = (-(len(scan) + 2)) % 16 + 16
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
    made.append((d, data, src, len(scan)))
plan, inp, out = B._cr2_batch(ctx, torch, [(m[0], m[1]) for m in made], W, H)
for run in range(2):
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    print(plan.results()[0], [bool(np.array_equal(B.gpu_frame(out, f, W, H), made[f][2])) for f in range(2)])
