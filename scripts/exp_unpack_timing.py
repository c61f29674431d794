"""GPU experiment: where does the host time of plan.run go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from rawspeed_amd import capi
import __graft_entry__ as ge
ge.build()
ctx = capi.Context(0)
F = 8
jobs = bench.unpack_jobs(F)
packed, px0 = bench.make_frames(1, 1)
inp = torch.from_numpy(np.tile(packed, F)).cuda()
h, opitch = bench.CFG2["h"], bench.out_pitch()
out = torch.empty(F * h * opitch, dtype=torch.uint8, device="cuda")
plan = ctx.unpack_plan(jobs)
s = torch.cuda.current_stream().cuda_stream
for timing in (False, True, False, True):
    plan.set_timing(timing)
    for _ in range(3): plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = []
    for _ in range(20):
        t1 = time.perf_counter(); plan.run(inp.data_ptr(), out.data_ptr(), s); ts.append(time.perf_counter() - t1)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("timing", timing, "enqueue %.3f ms/step" % (t_enq / 20 * 1e3), "total %.3f ms/step" % (t_all / 20 * 1e3),
          "max call %.3f ms" % (max(ts) * 1e3), "ktime", plan.kernel_time() if timing else None)
# single-frame latency
jobs1 = bench.unpack_jobs(1)
plan1 = ctx.unpack_plan(jobs1)
for _ in range(3): plan1.run(inp.data_ptr(), out.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): plan1.run(inp.data_ptr(), out.data_ptr(), s)
torch.cuda.synchronize()
print("single frame (L3-resident) %.1f us/frame" % ((time.perf_counter() - t0) / 50 * 1e6))
