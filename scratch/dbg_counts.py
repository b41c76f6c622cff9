import sys, numpy as np, torch, ctypes as C
sys.path.insert(0,'/root/repo')
import bench, imagemosaicing_amd as im
w,h,F=4000,3000,3
ws=3*w
A,g=bench.frame_layout(F,w,h,0)
ctx=im.Context(0)
frames=torch.empty((F,h*ws),dtype=torch.uint8,device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(),w,h,ws,A[k],0xC0FFEE,k,g[k],2.0)
for k in range(F):
    n=ctx.SiftExtractDev(k,frames[k].data_ptr(),w,h,ws,True)
    out=(C.c_int32*8)(); ctx.L.mi355_last_sift_counters(ctx._h,out); print(k,n,list(out))
kp,d=ctx.GetFeatures(0); print(np.unique(kp['octave']&255,return_counts=True), kp['response'][[0,-1]])
