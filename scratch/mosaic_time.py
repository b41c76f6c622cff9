"""canvas kernel timing on the C3 layout (ground-truth transforms): HIP events around mi355_mosaic_refined_dev"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import render_frames, ground_truth_h
F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
w, h = 4000, 3000
frames, A, g, ws = render_frames(ctx, torch, F, w, h)
h9 = np.stack([ground_truth_h(A, 0, k).reshape(9) for k in range(F)]).astype(np.float32)
h9[:, 6:8] = 0; h9[:, 8] = 1
cw, ch, cws, _ = im.mosaic_layout([w] * F, [h] * F, h9)
canvas = torch.empty(ch * cws, dtype=torch.uint8, device="cuda")
ptr = [frames[k].data_ptr() for k in range(F)]
for rep in range(3):
    ctx.profile_enable(True); ctx.profile_only("warp"); ctx.profile_reset()
    t0 = time.perf_counter(); ctx.MosaicImagesRefinedDev(ptr, [w] * F, [h] * F, [ws] * F, h9, canvas.data_ptr(), cw, ch, cws); ctx.synchronize(); dt = time.perf_counter() - t0
    ms, n, b = ctx.profile_get("warp")
    print("canvas %dx%d: kernel %.2f ms (%.1f us per frame, %.2f TB/s on the SURVEY figure 6P per frame), call %.2f ms" % (cw, ch, ms, ms * 1e3 / F, b / ms / 1e9, dt * 1e3))
