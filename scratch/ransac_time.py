"""C4-scale match+select+RANSAC timing: F frames (small, the pair stage does not depend on the frame size), window 182"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import render_frames
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = im.Context(0)
w, h = 1000, 750
frames, A, g, ws = render_frames(ctx, torch, F, w, h, per_row=25)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
pairs = im.pair_schedule(F, 182)
res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize()
t0 = time.perf_counter(); ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7); ctx.synchronize(); dt = time.perf_counter() - t0
ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 7)
ms = {c: round(ctx.profile_get(c)[0], 2) for c in ("match", "select", "ransac")}
r = res.cpu().numpy().view(im.PAIR_RESULT).reshape(-1)
print("fallback draws per pair: mean %.2f max %d" % (r["_pad"].mean(), r["_pad"].max()))
print("pairs", len(pairs), "wall %.1f ms" % (dt * 1e3), ms, "accepted", int(r["accepted"].sum()), "us/pair ransac %.2f" % (ms["ransac"] * 1e3 / len(pairs)),
      "=> C4 (74029 pairs) ransac %.3f s" % (ms["ransac"] * 74029 / len(pairs) / 1e3))
