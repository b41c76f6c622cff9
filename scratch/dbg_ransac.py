import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import imagemosaicing_amd as im
from tests.synth import synth_pairs
ctx=im.Context(0)
for n,of in [(396,0.35),(396,0.7),(200,0.5)]:
    p1,p2=synth_pairs(n,of,seed=5,size=(4000,3000))
    ctx.Ransac2D(p1,p2,2.5,1000,3)
    t=time.perf_counter(); r=ctx.Ransac2D(p1,p2,2.5,1000,3); dt=time.perf_counter()-t
    print(n,of,'ok',r[0],'inl',len(r[1]),'%.2f ms'%(dt*1e3))
# batch timing through match_pairs with synthetic features
from tests.test_gpu_parity import _synthetic_feature_pair
rng=np.random.default_rng(1)
P=64
for k in range(P):
    kp1,d1,kp2,d2=_synthetic_feature_pair(rng)
    ctx.SetFeatures(2*k,kp1,d1.astype(np.float32),4000,3000); ctx.SetFeatures(2*k+1,kp2,d2.astype(np.float32),4000,3000)
pairs=[(2*k,2*k+1) for k in range(P)]*8
res=ctx.MatchPairs(pairs,2.5,7)
t=time.perf_counter(); res=ctx.MatchPairs(pairs,2.5,7); dt=time.perf_counter()-t
print('match_pairs %d pairs: %.1f ms  (%.1f us/pair) accepted %d fallback draws total %d max %d'%(len(pairs),dt*1e3,dt*1e6/len(pairs),res['accepted'].sum(),res['_pad'].sum(),res['_pad'].max()))
