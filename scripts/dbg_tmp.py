import sys, os, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import bench_ljpeg as B
from rawspeed_amd import capi, abi, synth
import cases
ctx=capi.Context(0)
W,H=6720,4480
made=[]
for f in range(2):
    src=B.clipped_image(W,H,31+f)
    rows=cases.cr2_stream_from_image(src,2,W//2,H,cases.cr2_slices(3,2240,2240))
    scan,bits=synth.ljpeg_encode_scan(rows,2,[1<<13]*2,[B._nikon(),B._nikon()])
    d=abi.Cr2Desc(); d.n_comp,d.x_s_f,d.y_s_f=2,1,1; d.frame_w,d.frame_h=W//2,H
    d.num_slices,d.slice_width,d.last_slice_width=3,2240,2240
    abi.fill_recipe(d,synth.huff_tables(B._nikon()),[0,0],[1<<13]*2)
    pad=(-(len(scan)+2))%16+16
    data=np.concatenate([scan,np.array([0xFF,0xD9],np.uint8),np.zeros(pad,np.uint8)])
    made.append((d,data,src,len(scan)))
plan,inp,out=B._cr2_batch(ctx,torch,[(m[0],m[1]) for m in made],W,H)
s=torch.cuda.current_stream().cuda_stream
for i in range(3):
    plan.run(inp.data_ptr(),out.data_ptr(),s)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    rc,st,cons=plan.results()
    print("results ms", (time.perf_counter()-t0)*1e3, rc, file=sys.stderr)
