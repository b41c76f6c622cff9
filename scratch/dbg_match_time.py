import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, imagemosaicing_amd as im
w, h, F = 4000, 3000, 64
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
pairs = [(i, j) for i in range(F) for j in range(i + 1, min(F, i + 9))]
pairs = pairs[:int(sys.argv[1]) if len(sys.argv) > 1 else 499]
res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
for seed in (1, 2, 2, 3):
    t0 = time.perf_counter(); ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, seed); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print("seed %d: call %.2f ms, +sync %.2f ms" % (seed, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 3)
for c in ("match", "select", "ransac"): print(c, ctx.profile_get(c))
