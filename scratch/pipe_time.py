"""detect+describe rate under different stream layouts of the batch pipeline (a fresh ctx per layout: the knobs are read when a work area is made).
   python scratch/pipe_time.py [frames] [w] [h] [batch] [config ...]     config = name:opt=val,opt=val"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import imagemosaicing_amd as im
from tests.synth_survey import render_frames
F = int(sys.argv[1]) if len(sys.argv) > 1 else 96
w = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
h = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
cfgs = sys.argv[5:] or ["base:"]
ctx0 = im.Context(0)
frames, A, gains, ws = render_frames(ctx0, torch, F, w, h)
ctx0.close()
ref = None
for cfg in cfgs:
    name, _, opts = cfg.partition(":")
    ctx = im.Context(0)
    ctx.set_option("sift_batch", B)
    for o in filter(None, opts.split(",")):
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    def run():
        for k in range(F):
            ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
        ctx.synchronize()
    run()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); run(); run(); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e6 / (2 * F))
    kp, d = ctx.GetFeatures(F - 1)
    sig = (len(kp), int(d.astype(np.int64).sum()), float(kp["x"].astype(np.float64).sum()))
    if ref is None: ref = sig
    print("%-40s %7.1f us/frame (min of %s)  %s" % (name, min(ts), " ".join("%.1f" % t for t in ts), "same features" if sig == ref else "FEATURES DIFFER %s vs %s" % (sig, ref)), flush=True)
    ctx.close()
