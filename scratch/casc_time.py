# one batch of 8 frames 4000x3000 in flight: per-class event times of the pyramid, cascade on / off
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import imagemosaicing_amd as im
W, H, N = 4000, 3000, 8
g = torch.Generator(device="cpu"); g.manual_seed(1)
base = torch.randint(0, 255, (H // 8, W // 8, 3), generator=g, dtype=torch.uint8)
img = torch.nn.functional.interpolate(base.permute(2, 0, 1)[None].float(), size=(H, W), mode="bicubic")[0].permute(1, 2, 0).clamp(0, 255).to(torch.uint8).contiguous().cuda()
frames = [img.roll(17 * k, 1).contiguous() for k in range(N)]
CLS = ["cascade", "gauss_band", "gauss", "gauss_stream", "downsample", "extrema"]
for casc in ((2, 0, 2, 1) if not os.environ.get('CASC') else (int(os.environ['CASC']),) * 2):
    ctx = im.Context(0)
    ctx.set_option("sift_cascade", casc); ctx.set_option("sift_slots", 1)
    for rep in range(3):
        if rep == 1: ctx.profile_enable(True); ctx.profile_only(",".join(CLS)); ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.time()
        for k, f in enumerate(frames): ctx.SiftExtractDev(k, f.data_ptr(), W, H, W * 3)
        ctx.synchronize(); torch.cuda.synchronize()
        dt = time.time() - t0
    out = {c: ctx.profile_get(c) for c in CLS}
    print("cascade", casc, "wall ms/batch %.2f" % (dt * 1e3), {c: (round(v[0] / 2, 3), v[1] // 2) for c, v in out.items() if v[1]})
    ctx.close()
