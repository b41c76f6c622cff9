"""why a rank's detect+describe takes 17.4 ms in the proxy and 15.6 ms stand-alone: the same 63 frames, the proxy's surroundings added one by one"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, imagemosaicing_amd as im
w, h, F = 4000, 3000, 500
ws = 3 * w
A, g = bench.frame_layout(F, w, h, 0)
dev = torch.device("cuda", 0)
def make(use_stream):
    ctx = im.Context(0)
    if use_stream:
        st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
    return ctx
ctx = make(False)
frames = torch.empty((F, h * ws), dtype=torch.uint8, device=dev)
for k in range(F): ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC0FFEE, k, g[k], 2.0)
ctx.synchronize(); ctx.close()
def timeit(ctx, ids, reps=4):
    def run():
        for k in ids: ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
        ctx.synchronize()
    run(); run()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
for use_stream in (False, True):
    for ids_name, ids in (("first 63", list(range(63))), ("k = 0 mod 8", list(range(0, F, 8)))):
        for batch in (21, 32):
            ctx = make(use_stream); ctx.set_option("sift_batch", batch)
            print("caller stream %-5s frames %-12s batch %d: %.2f ms" % (use_stream, ids_name, batch, timeit(ctx, ids)), flush=True)
            ctx.close()
# the proxy's order of events: the whole survey once with batch 32, then the share with batch 21
ctx = make(True); ctx.set_option("sift_batch", 32)
for k in range(F): ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
ctx.synchronize()
ctx.set_option("sift_batch", 21)
print("after a whole-survey pass (500 feature sets resident), batch 21: %.2f ms" % timeit(ctx, list(range(0, F, 8))), flush=True)
from imagemosaicing_amd import dist as md
ex = md.Exchange(ctx, "rccl", strict=True)
print("with a one-rank RCCL communicator alive: %.2f ms" % timeit(ctx, list(range(0, F, 8))), flush=True)
own = list(range(0, F, 8))
ex.allgather_features(own, len(own), dev); ctx.synchronize()
print("after one feature all-gather: %.2f ms" % timeit(ctx, own), flush=True)
def step():
    for k in own: ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    ex.allgather_features(own, len(own), dev)
for _ in range(3): step()
ctx.synchronize()
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("detect + feature all-gather per step: %.2f ms" % min(ts), flush=True)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in own: ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    ctx.synchronize(); t1 = time.perf_counter(); ts.append((t1 - t0) * 1e3)
    ex.allgather_features(own, len(own), dev); ctx.synchronize()
print("detect alone inside such steps: %.2f ms" % min(ts), flush=True)
# the rest of a share step, added one by one
pairs = im.pair_schedule(F, 2, 0, 8)
res_r = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device=dev)
all_pairs = im.pair_schedule(F, 2)
results = torch.zeros((len(all_pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device=dev)
ctx.MatchPairsDev(all_pairs, results.data_ptr(), 2.5, 7)
r_all = ex.allgather_results(results, len(all_pairs), accepted_only=True)
wv, hv, wsv = [w] * F, [h] * F, [ws] * F
fptr = [frames[k].data_ptr() for k in range(F)]
label = im.select_connected_results(r_all, F); label[0] = 1
T = im.global_affine_align_results(r_all, F, fixed=[1 if (k == 0 or label[k] == 0) else 0 for k in range(F)], label=label)
h9 = T["m"].copy(); h9[label == 0, 8] = 0.0
cw, ch, cws, _ = im.mosaic_layout(wv, hv, h9)
canvas = torch.empty(int(1.2 * cws * ch) + (64 << 20), dtype=torch.uint8, device=dev)
def detect_ms(extra):
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in own: ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
        ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        extra(100 + i); ctx.synchronize()
    return min(ts[1:])
print("detect, then nothing: %.2f" % detect_ms(lambda s: None))
print("detect, then match of the rank's pairs: %.2f" % detect_ms(lambda s: ctx.MatchPairsDev(pairs, res_r.data_ptr(), 2.5, s)))
print("detect, then match + result all-gather: %.2f" % detect_ms(lambda s: (ctx.MatchPairsDev(pairs, res_r.data_ptr(), 2.5, s), ex.allgather_results(res_r, len(pairs), accepted_only=True))))
def full(s):
    ex.allgather_features(own, len(own), dev)
    ctx.MatchPairsDev(pairs, res_r.data_ptr(), 2.5, s); ex.allgather_results(res_r, len(pairs), accepted_only=True)
    T = im.global_affine_align_results(r_all, F, fixed=[1 if (k == 0 or label[k] == 0) else 0 for k in range(F)], label=label)
    ctx.MosaicImagesRefinedDev(fptr, wv, hv, wsv, h9, canvas.data_ptr(), cw, ch, cws, 0, ch // 8)
print("detect, then the whole rest of a share step: %.2f" % detect_ms(full))
def warp_only(s): ctx.MosaicImagesRefinedDev(fptr, wv, hv, wsv, h9, canvas.data_ptr(), cw, ch, cws, 0, ch // 8)
print("detect, then the canvas stripe only: %.2f" % detect_ms(warp_only))
def align_only(s): im.global_affine_align_results(r_all, F, fixed=[1 if (k == 0 or label[k] == 0) else 0 for k in range(F)], label=label)
print("detect, then the host alignment only: %.2f" % detect_ms(align_only))
