"""the pair stage of rank 0 of 8 (C3 strip, pairs i = 0 mod 8): which waves are slow (MI355_RANSAC_SPLIT_DBG=1)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, imagemosaicing_amd as im
w, h, F = 4000, 3000, 500
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
ctx = im.Context(0)
ctx.set_option("sift_batch", 32)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
for rk in (0, 7):
    pairs = im.pair_schedule(F, 2, rk, 8)
    res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
    ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
    ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7)
    ms = {c: round(ctx.profile_get(c)[0], 3) for c in ("match", "select", "ransac")}
    ctx.profile_enable(False)
    r = res.cpu().numpy().view(im.PAIR_RESULT).reshape(-1)
    print("rank", rk, len(pairs), "pairs: %.3f ms wall" % min(ts), ms, "inliers min %d" % r["n_in"].min(), "generic draws", int(r["_pad"].sum()), flush=True)
