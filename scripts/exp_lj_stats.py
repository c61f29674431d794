"""Round statistics of the LJPEG synchronisation on cfg 3 (experiment build with
-DRSX_EXPERIMENT; RSX_DEBUG=1 makes the plan print its per-stream counters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench_ljpeg as B
from rawspeed_amd import capi
ctx = capi.Context(0)
W, H = 6720, 4480
frames = int(os.environ.get("FRAMES", "8"))
if os.environ.get("WHAT", "cfg3") in ("cfg4mt", "cfg4"):  # (cfg 4, with a table per component)
    import numpy as np
    W, H = 8192, 5464
    try:
        src, jobs, datas, blobs, lens = B._dng_tiles(W, H, 4096, 2732, 2, 0,
                                                     two_tables=os.environ["WHAT"] == "cfg4mt")
    except TypeError:  # (an older bench_ljpeg)
        src, jobs, datas, blobs, lens = B._dng_tiles(W, H, 4096, 2732, 2, 0)
    inp = torch.from_numpy(np.concatenate(datas)).cuda()
    out = torch.zeros(B.out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
else:
    made = [B.make_cr2_frame(W, H, (3, 2240, 2240), seed=1 + f) for f in range(frames)]
    plan, inp, out = B._cr2_batch(ctx, torch, [(m[0], m[1]) for m in made], W, H)
plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
print(plan.results()[0])
