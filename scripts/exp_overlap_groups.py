"""Does a batch decode faster when the un-stuffing / guessing kernel of one half runs next
to the single-pass kernel of the other half (two plans, two streams, staggered)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench_ljpeg as B
from rawspeed_amd import capi
ctx = capi.Context(0)
G = int(os.environ.get("GROUPS", "2")); FR = int(os.environ.get("FRAMES", "128"))
per = FR // G
made = [B.make_cfg5_plan(ctx, torch, per, distinct=8, seed0=1000, first_frame=per * g) for g in range(G)]
whole = B.make_cfg5_plan(ctx, torch, FR, distinct=8, seed0=1000, first_frame=0)
streams = [torch.cuda.Stream() for _ in range(G)]
def run_whole():
    cur = torch.cuda.current_stream()
    streams[0].wait_stream(cur)
    whole[0].run(whole[1].data_ptr(), whole[2].data_ptr(), streams[0].cuda_stream)
    cur.wait_stream(streams[0])
def run_seq():
    cur = torch.cuda.current_stream()
    streams[0].wait_stream(cur)
    for p, i, o, m in made:
        p.run(i.data_ptr(), o.data_ptr(), streams[0].cuda_stream)
    cur.wait_stream(streams[0])
def run_par(delay_cycles):
    cur = torch.cuda.current_stream()
    for g, (p, i, o, m) in enumerate(made):
        st = streams[g]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            if g and delay_cycles:
                torch.cuda._sleep(int(delay_cycles * g))
            p.run(i.data_ptr(), o.data_ptr(), st.cuda_stream)
    for st in streams:
        cur.wait_stream(st)
def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("whole plan      %.3f ms" % timed(run_whole))
print("groups in turn  %.3f ms" % timed(run_seq))
print("groups at once  %.3f ms" % timed(lambda: run_par(0)))
for ms in (1.0, 2.0, 3.0, 4.5, 6.0):
    print("staggered %.1f ms  %.3f ms" % (ms, timed(lambda: run_par(ms * 1e-3 * 100e6))))
ok = all(B.check_cfg5(o, m, p.results()[2], per) for p, i, o, m in made)
print("exact", ok)
