import sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, bench_ljpeg as B
from rawspeed_amd import capi
ctx=capi.Context(0)
frames=int(sys.argv[1]) if len(sys.argv)>1 else 64
plan,inp,out,meta=B.make_cfg5_plan(ctx,torch,frames,distinct=8,seed0=100)
dt,kt,cons=B._time_plan(torch,plan,inp,out,5,2)
print("frames",frames,"ms",dt*1e3,"GPix/s",frames*8192*5464/dt/1e9)
print(B.LAST_KERNEL_TABLE)
