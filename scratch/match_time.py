"""bf_match_kernel alone: 64 frames of 4000x3000 resident, all pairs within a window of 40 (about 1800 pairs per call), exclusive
kernel time from the library's HIP events -> TFLOP/s of the exact all-pairs distance product (2 * 2000 * 2000 * 128 per pair)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, imagemosaicing_amd as im
w, h, F = 4000, 3000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
WIN = int(sys.argv[2]) if len(sys.argv) > 2 else 41
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
ctx = im.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device='cuda')
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
pairs = [(i, j) for i in range(F) for j in range(i + 1, min(F, i + WIN))]
res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device='cuda')
torch.cuda.synchronize()
ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 1); ctx.synchronize()
ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
for _ in range(3): ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 3)
ctx.synchronize()
n = 3 * len(pairs)
for c in ("match", "select", "ransac"):
    ms = ctx.profile_get(c)
    print(c, ms, "ms for", n, "pairs" + ("  -> %.0f TFLOP/s (%.3f of 2500)" % (n * 2.0 * 2000 * 2000 * 128 / 1e12 / (ms[0] / 1e3 if isinstance(ms, tuple) else ms / 1e3), n * 2.0 * 2000 * 2000 * 128 / 1e12 / ((ms[0] if isinstance(ms, tuple) else ms) / 1e3) / 2500) if c == "match" else ""))
r = res.cpu().numpy().view(im.PAIR_RESULT).reshape(-1)
print("accepted", int(r["accepted"].sum()), "checksum", int(r["n_in"].astype(np.int64).sum()), int(r["n_selected"].astype(np.int64).sum()))
print("resident workgroups per CU:", ctx.L.mi355_debug_match_occupancy())
