"""Where do the routes of a Nikon-type plan differ?  (debug aid of tests/test_gpu_nikon_routes.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_nikon_routes as T
import gpu_util, nikon_cases as N, golden_cases as G
from rawspeed_amd import synth
from oracle_lib import HostImage, Oracle
gpu = gpu_util.ctx()
oracle = Oracle()
bits, w, unc = 14, 2144, 1
for h, timing, pad, host in ((300, 1, 64, 0), (300, 0, 64, 0), (300, 0, 0, 0), (400, 0, 64, 0), (300, 0, 0, 1), (120, 0, 64, 0), (40, 0, 64, 0)):
    rng = np.random.default_rng([61, bits, w, h, unc])
    meta = N.metadata(70, 0, [2000, 2100, 2200, 2300])
    P = N.parse(meta, bits, h)
    src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1)
    pu = P["p_up"]
    data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]], synth.NIKON_TREE[P["huff_select"]])
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    d = N.desc(P, bits, bool(unc))
    want = HostImage(w, h)
    assert oracle.nikon(d, data, want) == 0
    os.environ["RSX_NO_FAST_NK"] = "1"
    if host:
        img = HostImage(w, h)
        st = gpu.nikon_decompress(d, data, img.view())
        print("host call", h, st, "equal:", np.array_equal(img.u16(), want.u16()))
        continue
    in_host = np.concatenate([data, np.zeros(pad, np.uint8)])
    jobs = [T._nikon_job(gpu_util, d, data, w, h, want.pitch, 0, 0)]
    plan = gpu.nikon_plan(jobs)
    os.environ.pop("RSX_NO_FAST_NK")
    d_in = gpu_util.to_dev(in_host)
    for run in range(3):
        d_out = torch.full((want.buf.size + 16,), 0xA5, dtype=torch.uint8, device="cuda")
        if timing:
            plan.set_timing(True)
        plan.run(d_in.data_ptr(), d_out.data_ptr())
        rc, status, _ = plan.results()
        names = []
        if timing:
            tab = plan.kernel_table()
            names = [n for n, _ in tab[0]] if tab else []
        g = d_out.cpu().numpy()[:want.buf.size].view(np.uint16).reshape(h, -1)[:, :w]
        bad = np.argwhere(g != want.u16()[:, :w])
        print("h", h, "timing", timing, "pad", pad, "run", run, "status", status, "bad px", len(bad),
              (tuple(bad[0]), tuple(bad[-1])) if len(bad) else "", names)
    plan.close()
