import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import frame_layout
w, h, F = 4000, 3000, 96
ws = 3 * w
A, g = frame_layout(F, w, h, 0)
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((8, h * ws), dtype=torch.uint8, device='cuda')
for k in range(8): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
ctx.synchronize()
host = [frames[k].cpu().numpy().reshape(h, w, 3).copy() for k in range(8)]
pinned = [torch.from_numpy(x).pin_memory().numpy() for x in host]
for name, src in (("pageable", host), ("pinned", pinned)):
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(F): ctx.SiftExtractHost(k, src[k % 8])
        ctx.synchronize(); dt = time.perf_counter() - t0
    print("%s host frames: %.1f frames/s (%.2f ms/frame, %.1f GB/s)" % (name, F / dt, dt / F * 1e3, F * h * ws / dt / 1e9))
