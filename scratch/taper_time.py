"""detect+describe of a rank's 63 frames under different batch plans (option sift_flush closes a batch early)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import imagemosaicing_amd as im
from tests.synth_survey import render_frames
F, w, h = 63, 4000, 3000
ctx0 = im.Context(0)
frames, A, gains, ws = render_frames(ctx0, torch, F, w, h)
ctx0.close()
plans = [("21,21,21", [21, 21, 21]), ("32,31", [32, 31]), ("32,23,8", [32, 23, 8]), ("32,16,15", [32, 16, 15]), ("28,20,10,5", [28, 20, 10, 5]), ("32,16,8,7", [32, 16, 8, 7]),
         ("24,16,12,8,3", [24, 16, 12, 8, 3]), ("16,16,16,15", [16, 16, 16, 15]), ("32,20,11", [32, 20, 11]), ("26,22,15", [26, 22, 15])]
for opts in ({}, {"serial_heavy": 1}, {"sift_slots": 4}, {"sift_slots": 4, "serial_heavy": 1}, {"sift_slots": 2}):
    for name, plan in plans:
        ctx = im.Context(0)
        ctx.set_option("sift_batch", 32)
        for k, v in opts.items(): ctx.set_option(k, v)
        bounds = set(np.cumsum(plan).tolist())
        def run():
            for k in range(F):
                ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
                if (k + 1) in bounds: ctx.set_option("sift_flush", 1)
            ctx.synchronize()
        run(); run()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%-28s plan %-16s %6.2f ms (%.1f us/frame)  [%s]" % (opts, name, min(ts), min(ts) * 1e3 / F, " ".join("%.2f" % t for t in ts)), flush=True)
        ctx.close()
