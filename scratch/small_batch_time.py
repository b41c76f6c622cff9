"""latency of the pair stage for few pairs: adjacent pairs (accepted: the closing refinement runs) and a single mi355_ransac2d call, with the
   split form (several workgroups per pair) on and off.   python scratch/small_batch_time.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, imagemosaicing_amd as im
from tests.synth import synth_pairs
w, h, F = 4000, 3000, 64
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
adj = [(i, i + 1) for i in range(F - 1)]
ref = {}
for S in (0, -1, 4, 8):
    ctx.set_option("ransac_split", S)
    for npairs in (1, 8, 63):
        pairs = adj[:npairs]
        res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
        ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
        ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7)
        ms = {c: round(ctx.profile_get(c)[0], 3) for c in ("match", "select", "ransac")}
        ctx.profile_enable(False)
        b = res.cpu().numpy().tobytes()
        same = ref.setdefault(npairs, b) == b
        r = np.frombuffer(b, im.PAIR_RESULT)
        print("ransac_split %2d  %3d adjacent pairs: %.3f ms wall (min of 5), kernels %s, accepted %d, %s" % (S, npairs, min(ts), ms, int(r["accepted"].sum()), "same records" if same else "RECORDS DIFFER"), flush=True)
    p1, p2 = synth_pairs(396, 0.4, seed=9, size=(4000, 3000))
    ctx.Ransac2D(p1, p2, 2.5, 1000, 7)
    ts = []
    for rep in range(10):
        t0 = time.perf_counter(); out = ctx.Ransac2D(p1, p2, 2.5, 1000, 7); ts.append((time.perf_counter() - t0) * 1e3)
    print("ransac_split %2d  one mi355_ransac2d call (396 correspondences, 40 %% outliers, %d inliers): %.3f ms (min of 10; median %.3f)" % (S, len(out[1]), min(ts), sorted(ts)[5]), flush=True)
