"""per-class kernel time of detect+describe (HIP events, one batch work area in flight = exclusive durations) and the pipelined rate.
   python scratch/sift_time.py [frames] [w] [h] [batch]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import imagemosaicing_amd as im
from tests.synth_survey import render_frames
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
h = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
SERIAL = len(sys.argv) > 5 and sys.argv[5] == "serial"      # one slot, no event brackets: for a rocprofv3 kernel trace with exclusive durations
ctx = im.Context(0)
ctx.set_option("sift_batch", B)
frames, A, gains, ws = render_frames(ctx, torch, F, w, h)
CLS = ("gauss_stream", "gauss", "downsample", "extrema", "refine", "kp_select", "orient", "topk", "describe", "features")
def run():
    for k in range(F):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    ctx.synchronize()
run()
if SERIAL:
    ctx.set_option("sift_slots", 1)
    t0 = time.perf_counter(); run(); run(); t1 = time.perf_counter()
    print("serial pass (one batch work area in flight, no event brackets): %d frames %dx%d batch %d, %.1f us/frame" % (2 * F, w, h, B, (t1 - t0) * 1e6 / (2 * F)))
    sys.exit(0)
ctx.set_option("sift_slots", 1)
ctx.profile_enable(True); ctx.profile_only(None); ctx.profile_reset()
t0 = time.perf_counter(); run(); t1 = time.perf_counter()
tot = 0
for c in CLS:
    ms, n, b = ctx.profile_get(c)
    tot += ms
    print("%-14s %8.1f us/frame  %6d launches  %7.1f us/launch  %6.2f TB/s alg" % (c, ms * 1e3 / F, n, ms * 1e3 / max(n, 1), (b / 1e12) / (ms / 1e3) if ms else 0))
print("sum of classes %.1f us/frame; 1 slot wall (events on) %.1f us/frame" % (tot * 1e3 / F, (t1 - t0) * 1e6 / F))
ctx.profile_enable(False)
for slots in (1, 2, 3, 4):
    ctx.set_option("sift_slots", slots)
    run()
    t0 = time.perf_counter(); run(); run(); t1 = time.perf_counter()
    print("slots %d: %.1f us/frame" % (slots, (t1 - t0) * 1e6 / (2 * F)))
print("counters of the last frame (extrema, refined, keypoints, kept, overflow):", ctx.last_sift_counters()[:5] if hasattr(ctx, "last_sift_counters") else "")
