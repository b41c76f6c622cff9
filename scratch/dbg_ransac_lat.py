import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, imagemosaicing_amd as im
w, h, F = 4000, 3000, 40
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
allp = [(i, j) for i in range(F) for j in range(i + 1, F)]
for npairs in (1, 8, 64, 256, 512, 780):
    pairs = allp[:npairs]
    res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
    ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
    ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7)
    ms = {c: ctx.profile_get(c)[0] for c in ("match", "select", "ransac")}
    ctx.profile_enable(False)
    r = res.cpu().numpy().view(im.PAIR_RESULT).reshape(-1)
    print(npairs, {k: round(v, 3) for k, v in ms.items()}, "accepted", int(r["accepted"].sum()), "nsel", int(r["n_selected"].mean()), "fallback draws", int(r["_pad"].sum()) if "_pad" in r.dtype.names else "?")
