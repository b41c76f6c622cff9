"""randomised parity soak of match + select + Ransac2D: GPU vs oracle on synthetic feature pairs
    python scratch/soak_pairs.py SEED SECONDS [large]     large: keep-all sized images, 2049 .. 32768 keypoints on either side (match.hip's large-pair path)"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.test_gpu_parity import _synthetic_feature_pair, bits
o = oracle_lib.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
large = len(sys.argv) > 3 and sys.argv[3] == "large"
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    nk = int(rng.integers(40, 2049)) if not large else int(min(32768, 2049 + rng.exponential(3000))); ov = float(rng.uniform(0.0, 0.9)); w = int(rng.integers(300, 4001)); h = int(rng.integers(300, 3001))
    kp1, d1, kp2, d2 = _synthetic_feature_pair(rng, n=nk, w=w, h=h, overlap=ov)
    if large and rng.random() < 0.5:                      # ragged: one side short (<= 2048) or differently long
        m = int(rng.integers(40, nk)); kp2, d2 = kp2[:m], d2[:m]
    seed = int(rng.integers(1, 1 << 31)); dist = float(rng.choice([1.0, 2.5, 4.0]))
    c.SetFeatures(0, kp1, d1.astype(np.float32), w, h); c.SetFeatures(1, kp2, d2.astype(np.float32), w, h)
    r = c.MatchPairs([(0, 1)], dist, seed)[0]
    xy1 = np.stack([kp1["x"], kp1["y"]], 1); xy2 = np.stack([kp2["x"], kp2["y"]], 1)
    nin, i1, i2, H, ns = o.match_pair(xy1, d1, xy2, d2, w, h, dist, seed)
    n_in = int(r["n_in"])
    ok = int(r["n_selected"]) == ns and (n_in if n_in > 30 else 0) == nin
    if ok and nin: ok = np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["b"][:nin], i2[:nin]) and np.array_equal(bits(r["H"]), bits(H))
    n += 1
    if not ok: bad += 1; print("MISMATCH", nk, ov, w, h, seed, dist, n_in, nin, int(r["n_selected"]), ns, flush=True)
print("pair soak%s: %d pairs, %d mismatches, %.0f s" % (" (large)" if large else "", n, bad, time.time() - t0))
