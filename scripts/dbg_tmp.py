import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import bench_ljpeg as B
import __graft_entry__ as ge
ge.build()
from rawspeed_amd import capi
ctx=capi.Context(0)
for (W,H,tw,th,ri) in ((1024,64,512,32,8),(8192,5464,4096,2732,683),(8192,5464,4096,2732,1366)):
    src, jobs, datas, blobs, lens = B._dng_tiles(W,H,tw,th,2, rows_per_ri=ri)
    inp=torch.from_numpy(np.concatenate(datas)).cuda()
    out=torch.zeros(B.out_pitch(W)*H,dtype=torch.uint8,device='cuda')
    plan=ctx.ljpeg_plan(jobs)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc,st,cons=plan.results()
    got=B.gpu_frame(out,0,W,H)
    bad=np.argwhere(got!=src)
    print(W,H,ri,"rc",rc,st,"cons",cons,lens,"mismatch",len(bad), bad[:3].tolist())
