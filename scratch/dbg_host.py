import sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0,'/root/repo')
import bench, imagemosaicing_amd as im
w,h,F=4000,3000,24
ws=3*w
A,g=bench.frame_layout(F,w,h,0)
ctx=im.Context(0)
st=torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames=torch.empty((F,h*ws),dtype=torch.uint8,device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(),w,h,ws,A[k],0xC0FFEE,k,g[k],2.0)
for rep in range(3):
    ctx.synchronize(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for k in range(F): ctx.SiftExtractDev(k,frames[k].data_ptr(),w,h,ws)
    t1=time.perf_counter()
    ctx.synchronize(); torch.cuda.synchronize()
    t2=time.perf_counter()
    print('enqueue %.2f ms/frame, total %.2f ms/frame'%((t1-t0)/F*1e3,(t2-t0)/F*1e3))
