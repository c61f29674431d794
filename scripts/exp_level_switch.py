"""The scenario of tests/test_gpu_fast_path.py::test_lds_level_follows_the_data_of_a_reused_plan
with the experiment build's per-stream statistics (RSX_LIB=<stats build> RSX_DEBUG=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench_ljpeg as B
import cases as C
from rawspeed_amd import abi, capi, synth
gpu = capi.Context(0)
W, H = 4480, 1024


def encode(src):
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, C.cr2_slices(2, W // 2, W // 2))
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()])
    return scan


noisy = synth.sensor_image(W, H, 14, seed=90)
dense = B.clipped_image(W, H, 91)
sa, sb = encode(noisy), encode(dense)
n = len(sa) + 2 + ((-(len(sa) + 2)) % 16 + 16)


def blob(scan):
    out = np.zeros(n, np.uint8)
    out[:len(scan)] = scan
    out[len(scan):len(scan) + 2] = (0xFF, 0xD9)
    return out


d = abi.Cr2Desc()
d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
d.frame_w, d.frame_h = W // 2, H
d.num_slices, d.slice_width, d.last_slice_width = 2, W // 2, W // 2
abi.fill_recipe(d, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
plan, inp, out = B._cr2_batch(gpu, torch, [(d, blob(sa))], W, H)
s = torch.cuda.current_stream().cuda_stream
plan.set_timing(True)
for name, scan, src in (("noisy", sa, noisy), ("noisy", sa, noisy), ("dense", sb, dense),
                        ("dense", sb, dense), ("dense", sb, dense), ("noisy", sa, noisy)):
    inp.copy_(torch.from_numpy(blob(scan)))
    out.zero_()
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    rc, st, cons = plan.results()
    ok = np.array_equal(B.gpu_frame(out, 0, W, H), src)
    tab = plan.kernel_table()
    print("RUN", name, rc, list(cons) == [len(scan)], ok, [k for k, _ in tab[0]] if tab else [], flush=True)
