"""randomised parity soak of the SURF variant: GPU extraction vs oracle_surf.c over random sizes / contents / thresholds / limits (incl. more than
8192 keypoints per image), and the pair stage (exact 1-NN, threshold walk, Ransac) on random pairs of the extracted images"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth_frames import terrain, strip
o = oracle_lib.load_oracle_fast()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n = 0; bad = 0; npair = 0; badp = 0; big = 0
c = im.Context(0)
while time.time() - t0 < budget:
    kind = int(rng.integers(0, 4))
    if kind == 0:                                        # an overlapping pair of a strip: the pair stage accepts
        w, h = int(rng.integers(300, 900)), int(rng.integers(240, 700)); imgs = strip(2, w, h, seed=int(rng.integers(1 << 30)))[0]
    elif kind == 1:                                      # many keypoints
        w, h = int(rng.integers(1200, 2100)), int(rng.integers(900, 1500)); imgs = [terrain(w, h, seed=int(rng.integers(1 << 30)))]
    elif kind == 2:
        w, h = int(rng.integers(16, 400)), int(rng.integers(16, 300)); imgs = [terrain(w, h, seed=int(rng.integers(1 << 30))) for _ in range(2)]
    else:
        w, h = int(rng.integers(64, 500)), int(rng.integers(64, 400)); imgs = [np.random.default_rng(int(rng.integers(1 << 30))).integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    thr = float(rng.choice([0.5, 2.0, 10.0, 50.0, 400.0, 3000.0]))
    mk = int(rng.choice([100, 3000, 8192, 8193, 20000, 32768]))
    feats = []
    for k, img in enumerate(imgs):
        kp, d = c.SurfExtract(k, img, thr, mk)
        okp, od = o.surf(img, thr, mk)
        ok = len(kp) == len(okp) and all(np.array_equal(kp[f].view(np.uint32) if kp[f].dtype.kind == "f" else kp[f], okp[f].view(np.uint32) if okp[f].dtype.kind == "f" else okp[f])
                                         for f in ("x", "y", "size", "angle", "response", "octave", "class_id")) and np.array_equal(d.view(np.uint32), od.view(np.uint32))
        n += 1; big += int(len(okp) > 8192)
        if not ok: bad += 1; print("MISMATCH extract", img.shape, thr, mk, len(kp), len(okp), flush=True)
        feats.append((okp, od))
    if len(imgs) == 2 and len(feats[0][0]) <= 6000 and len(feats[1][0]) > 0 and len(feats[0][0]) > 0:
        md = float(rng.choice([0.05, 0.12, 0.3, 0.5, 2.5])); mf = int(rng.choice([1, 30, 150, 200, 400])); seed = int(rng.integers(1, 1000))
        r = c.SurfMatchPairs(np.array([[0, 1]], np.int32), 2.5, seed, match_dist=md, max_features=mf)[0]
        nin, i1, i2, Ho, ns = o.surf_match_pair(feats[0], feats[1], 2.5, seed, md, mf)
        ok = int(r["n_selected"]) == ns and int(r["accepted"]) == int(nin > 18)
        if ok and nin > 18: ok = int(r["n_in"]) == nin and np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["b"][:nin], i2[:nin]) and np.array_equal(r["H"].view(np.uint32), Ho.view(np.uint32))
        npair += 1
        if not ok: badp += 1; print("MISMATCH pair", imgs[0].shape, thr, mk, md, mf, seed, int(r["n_selected"]), ns, int(r["n_in"]), nin, flush=True)
c.close()
print("surf soak: %d images (%d with more than 8192 keypoints), %d mismatches; %d pairs, %d mismatches, %.0f s" % (n, big, bad, npair, badp, time.time() - t0))
