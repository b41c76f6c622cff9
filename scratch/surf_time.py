"""SURF variant timing on a 12 MP frame and on the C2 strip pair stage"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import render_frames, host_image
ctx = im.Context(0)
w, h, F = 4000, 3000, 8
frames, A, g, ws = render_frames(ctx, torch, F, w, h)
imgs = [host_image(frames, k, w, h, ws) for k in range(F)]
for thr in (50.0, 400.0):
    ctx.SurfExtract(0, imgs[0], thr, 8192)
    t0 = time.perf_counter()
    for k in range(F): kp, d = ctx.SurfExtract(k, imgs[k], thr, 8192)
    dt = (time.perf_counter() - t0) / F
    ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
    ctx.SurfExtract(0, imgs[0], thr, 8192)
    ms = {c: round(ctx.profile_get(c)[0], 2) for c in ("surf_integral", "surf_det", "surf_sort", "surf_describe")}
    ctx.profile_enable(False)
    print("hessian %.0f: %d keypoints, %.1f ms per 12 MP frame incl. 36 MB upload + feature download; kernels %s" % (thr, len(kp), dt * 1e3, ms))
pairs = im.surf_pair_schedule(F)
t0 = time.perf_counter(); r = ctx.SurfMatchPairs(pairs, 2.5, 3); dt = time.perf_counter() - t0
print("%d ring pairs (8192 x 8192 float matching each): %.1f ms, accepted %d" % (len(pairs), dt * 1e3, int(r["accepted"].sum())))
