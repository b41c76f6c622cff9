"""randomised parity soak of the SURF path: GPU vs oracle_surf.c over random sizes / contents / thresholds, features and pair records"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth_frames import terrain, strip
o = oracle_lib.load_oracle_fast()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0; npairs = 0
def content(w, h, kind, seed):
    if kind == 0: return terrain(w, h, seed=seed)
    r = np.random.default_rng(seed)
    if kind == 1: return r.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = terrain(w, h, seed=seed); img[: h // 2] = 90; return img
while time.time() - t0 < budget:
    w = int(rng.integers(40, 1300)); h = int(rng.integers(40, 900))
    kind = int(rng.integers(0, 3)); thr = float(rng.choice([20.0, 50.0, 400.0, 3000.0])); mk = int(rng.choice([64, 500, 3000, 8192]))
    img = content(w, h, kind, int(rng.integers(1 << 30)))
    kp, d = c.SurfExtract(0, img, thr, mk)
    okp, od = o.surf(img, thr, mk)
    ok = len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    n += 1
    if not ok: bad += 1; print("SURF MISMATCH", w, h, kind, thr, mk, len(kp), len(okp), flush=True)
    if rng.random() < 0.3:
        frames, _ = strip(2, int(rng.integers(300, 700)), int(rng.integers(240, 500)), seed=int(rng.integers(1 << 20)), overlap=float(rng.uniform(0.3, 0.8)))
        f = []
        for q, fr in enumerate(frames):
            c.SurfExtract(10 + q, fr, 50.0, 2000); f.append(o.surf(fr, 50.0, 2000))
        seed = int(rng.integers(1, 1 << 20))
        r = c.SurfMatchPairs([(10, 11), (11, 10)], 2.5, seed)
        for p, (i, j) in enumerate([(0, 1), (1, 0)]):
            nin, i1, i2, Ho, ns = o.surf_match_pair(f[i], f[j], 2.5, seed)
            okp = int(r[p]["n_selected"]) == ns and int(r[p]["accepted"]) == int(nin > 18)
            if okp and nin > 18:
                okp = int(r[p]["n_in"]) == nin and np.array_equal(r[p]["a"][:nin], i1[:nin]) and np.array_equal(r[p]["H"].view(np.uint32), Ho.view(np.uint32))
            npairs += 1
            if not okp: bad += 1; print("SURF PAIR MISMATCH", frames[0].shape, seed, flush=True)
print("surf soak: %d images, %d pairs, %d mismatches, %.0f s" % (n, npairs, bad, time.time() - t0))
