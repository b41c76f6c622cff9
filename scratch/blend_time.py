"""LaplacianPyramidBlending in one call (mi355_mosaic_blended): default the C2 strip (50 frames 1920x1080, ground-truth transforms);
   blend_time.py 100 4000 3000 10 = a 10 x 10 block of 12 MP frames (C5-sized canvas)"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import render_frames, host_image, ground_truth_h
F = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = im.Context(0)
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920; h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
per_row = int(sys.argv[4]) if len(sys.argv) > 4 else F
frames, A, g, ws = render_frames(ctx, torch, F, w, h, per_row=per_row)
imgs = [host_image(frames, k, w, h, ws) for k in range(F)]
h9 = np.stack([ground_truth_h(A, 0, k).reshape(9) for k in range(F)]).astype(np.float32)
keep = im.resample_by_overlap([w] * F, [h] * F, h9, 0.7)
for rep in range(3):
    t0 = time.perf_counter(); out, ow, oh, ows = ctx.MosaicBlended(imgs, h9, keep=keep, band=5); dt = time.perf_counter() - t0
    print("blend %d frames (kept %d) -> %dx%d canvas: %.1f ms wall (incl. %d MB upload + %d MB download)" % (F, int(keep.sum()), ow, oh, dt * 1e3, F * w * h * 3 >> 20, ows * oh >> 20))
ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
ctx.MosaicBlended(imgs, h9, keep=keep, band=5)
print({c: round(ctx.profile_get(c)[0], 2) for c in ("warp", "distmap", "owner")})
t0 = time.perf_counter(); c2, cw, ch, cws = ctx.MosaicImagesRefined(imgs, h9); print("no-blend canvas: %.1f ms wall" % ((time.perf_counter() - t0) * 1e3))
