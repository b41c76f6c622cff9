"""which pairs of a rank's 63 make the pair stage slow: every adjacent pair on its own, both RANSAC forms"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, imagemosaicing_amd as im
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
mode = sys.argv[1] if len(sys.argv) > 1 else "each"
if mode == "trace":      # for rocprofv3 --kernel-trace: a few calls of the 63-pair stage in each form
    res = torch.zeros((63, im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
    for S in (0, 8):
        ctx.set_option("ransac_split", S)
        for rep in range(4):
            ctx.MatchPairsDev(adj, res.data_ptr(), 2.5, 7); ctx.synchronize()
    sys.exit(0)
res = torch.zeros((1, im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
rows = []
for p in adj:
    t = {}
    for S in (0, 8):
        ctx.set_option("ransac_split", S)
        ctx.MatchPairsDev([p], res.data_ptr(), 2.5, 7); ctx.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); ctx.MatchPairsDev([p], res.data_ptr(), 2.5, 7); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        t[S] = min(ts)
    r = res.cpu().numpy().view(im.PAIR_RESULT).reshape(-1)[0]
    rows.append((p, t[0], t[8], int(r["n_selected"]), int(r["n_in"]), int(r["_pad"])))
for p, a, b, ns, ni, fb in rows:
    print("pair %s  one workgroup %.3f ms  split %.3f ms  selected %d inliers %d generic draws %d" % (p, a, b, ns, ni, fb))
