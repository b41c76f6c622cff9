import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import bench, imagemosaicing_amd as im
w,h,F=4000,3000,40
ws=3*w
A,g=bench.frame_layout(F,w,h,0)
ctx=im.Context(0)
frames=torch.empty((F,h*ws),dtype=torch.uint8,device='cuda')
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(),w,h,ws,A[k],0xC0FFEE,k,g[k],2.0)
for k in range(F): n=ctx.SiftExtractDev(k,frames[k].data_ptr(),w,h,ws)
pairs=im.pair_schedule(F,2)
res=ctx.MatchPairs(pairs,2.5,101)
corners=np.array([[0,0,1],[w-1,0,1],[w-1,h-1,1],[0,h-1,1]],np.float64).T
for r in res:
    i,j=int(r['i']),int(r['j'])
    Hg=np.linalg.inv(bench.affine3(A[i]))@bench.affine3(A[j])
    He=r['H'].astype(np.float64).copy(); He[8]=1; He=He.reshape(3,3)
    a,b=He@corners,Hg@corners
    e=np.abs(a[:2]/a[2]-b[:2]/b[2]).max()
    if e>0.2 or r['n_in']<200: print(i,j,int(r['n_selected']),int(r['n_in']),int(r['accepted']),'err %.3f'%e, 'H8 %.3f'%r['H'][8], Hg[0,2].round(1),Hg[1,2].round(1))
