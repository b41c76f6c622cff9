"""mi355_mosaic_blended_dev on a dense block of resident 12 MP frames (C5-like geometry, F frames): wall time of the second call, checksum"""
import sys, time, hashlib, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests.synth_survey import affine3, block_layout
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, h = 4000, 3000
ws = (3 * w + 3) & ~3
ctx = im.Context(0)
A = block_layout(F, w, h, cols=cols, extent=20000.0 * cols / 50.0 + 4000)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device="cuda")
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC5C5C5, k, 1.0, 2.0)
ctx.synchronize()
A0i = np.linalg.inv(affine3(A[0]))
h9 = np.stack([(A0i @ affine3(A[k])).reshape(9) for k in range(F)]).astype(np.float32)
fptr = [frames[k].data_ptr() for k in range(F)]
wv, hv, wsv = [w] * F, [h] * F, [ws] * F
keep = im.resample_by_overlap(wv, hv, h9, 0.7)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, bw, bh, bws = ctx.MosaicBlendedDev(fptr, wv, hv, wsv, h9, keep=keep, band=5)
    ctx.synchronize(); dt = time.perf_counter() - t0
    print("blend %d chips -> %d x %d: %.1f ms (%.2f ms per chip)" % (int(keep.sum()), bw, bh, dt * 1e3, dt * 1e3 / F))
print("sha", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16])
