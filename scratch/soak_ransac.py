"""randomised parity soak of Ransac2D alone: GPU vs oracle, including degenerate inputs (duplicated points, collinear
points, integer coordinates, tiny sets) that drive the 4-point solve into failed inversions and pivot searches"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib as ol
o = ol.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0; acc = 0
while time.time() - t0 < budget:
    m = int(rng.integers(4, 401)); kind = int(rng.integers(0, 6))
    w, h = 4000.0, 3000.0
    p1 = np.zeros(m, ol.SFPOINT); p2 = np.zeros(m, ol.SFPOINT)
    p2['x'] = rng.uniform(0, w, m); p2['y'] = rng.uniform(0, h, m)
    A = np.array([[1 + rng.normal(0, .02), rng.normal(0, .02), rng.uniform(-800, 800)], [rng.normal(0, .02), 1 + rng.normal(0, .02), rng.uniform(-600, 600)], [rng.normal(0, 1e-6), rng.normal(0, 1e-6), 1]])
    q = A @ np.stack([p2['x'], p2['y'], np.ones(m)]); p1['x'] = q[0] / q[2] + rng.normal(0, 0.4, m); p1['y'] = q[1] / q[2] + rng.normal(0, 0.4, m)
    out = rng.random(m) < rng.uniform(0, 0.9)
    p1['x'][out] = rng.uniform(0, w, out.sum()); p1['y'][out] = rng.uniform(0, h, out.sum())
    if kind == 1:      # duplicated source points (one train keypoint matched by several queries)
        k = rng.integers(0, m, m // 2); p2['x'][: m // 2] = p2['x'][k]; p2['y'][: m // 2] = p2['y'][k]
    elif kind == 2:    # duplicated pairs
        k = rng.integers(0, m, m // 3); p2[: m // 3] = p2[k]; p1[: m // 3] = p1[k]
    elif kind == 3:    # integer coordinates, many collinear
        p2['x'] = np.round(p2['x'] / 50) * 50; p2['y'] = np.round(p2['y'] / 50) * 50; p1['x'] = np.round(p1['x']); p1['y'] = np.round(p1['y'])
    elif kind == 4:    # all sources on one line
        p2['y'] = 0.5 * p2['x'] + 10
    elif kind == 5:    # pure garbage
        p1['x'] = rng.uniform(0, w, m); p1['y'] = rng.uniform(0, h, m)
    seed = int(rng.integers(1, 1 << 31)); dist = float(rng.choice([1.0, 2.5, 4.0])); st = int(rng.choice([1000, 1000, 200, 37, 1, 5, 256, 257, 1700, 4999]))
    ok, i1, i2, H = c.Ransac2D(p1, p2, dist, st, seed)
    ok2, j1, j2, H2 = o.ransac2d(p1, p2, dist, st, seed)
    good = (ok == ok2) and len(i1) == len(j1) and np.array_equal(i1.view(np.uint8), j1.view(np.uint8)) and np.array_equal(i2.view(np.uint8), j2.view(np.uint8)) and np.array_equal(H.view(np.uint32), H2.view(np.uint32))
    n += 1; acc += int(ok2 != 0)
    if not good:
        import os; os.makedirs("gpurun_out/soak", exist_ok=True); np.savez("gpurun_out/soak/ransac_case_%d.npz" % n, p1=p1, p2=p2, dist=dist, st=st, seed=seed, gi1=i1, gi2=i2, gH=H, ok=ok)
    if not good: bad += 1; print("MISMATCH kind", kind, "m", m, "seed", seed, dist, st, ok, ok2, len(i1), len(j1), flush=True)
print("ransac soak: %d cases (%d ok), %d mismatches, %.0f s" % (n, acc, bad, time.time() - t0))
