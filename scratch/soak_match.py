"""randomised parity soak of the int8 brute-force matcher: GPU (mi355_bf_match: index, d2, second d2) vs the CPU integer brute force,
ragged sizes, low-entropy descriptors (many exact ties), duplicated rows, the ends of the u8 range"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
o = oracle_lib.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0
def desc(nr, kind):
    if kind == 0: return rng.integers(0, 256, (nr, 128)).astype(np.uint8)
    if kind == 1: return rng.choice(np.array([0, 1, 255], np.uint8), (nr, 128))                     # extreme values, many ties
    if kind == 2: return (rng.integers(0, 3, (nr, 128)) * rng.integers(0, 2, (nr, 1))).astype(np.uint8)    # half the rows all zero
    if kind == 3: d = rng.integers(0, 256, (max(1, nr // 7), 128)).astype(np.uint8); return d[rng.integers(0, len(d), nr)]   # heavy duplication
    v = rng.gamma(0.6, 30.0, (nr, 128)); v = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-9) * 512.0           # SIFT-like
    return np.clip(np.round(v), 0, 255).astype(np.uint8)
while time.time() - t0 < budget:
    n1 = int(rng.choice([rng.integers(1, 2049), rng.integers(1, 70), 2048, 2000, 512, 513, 64, 65]))
    n2 = int(rng.choice([rng.integers(1, 2049), rng.integers(1, 70), 2048, 2000, 32, 33, 31, 1]))
    kind = int(rng.integers(0, 5))
    d1, d2 = desc(n1, kind), desc(n2, int(rng.integers(0, 5)) if rng.random() < 0.3 else kind)
    if n1 > 3 and n2 > 3 and rng.random() < 0.5: d1[: min(n1, n2) // 2] = d2[: min(n1, n2) // 2]   # exact matches at distance 0
    kp1 = np.zeros(n1, im.KEYPOINT); kp2 = np.zeros(n2, im.KEYPOINT)
    kp1["x"] = rng.uniform(5, 995, n1); kp1["y"] = rng.uniform(5, 745, n1); kp2["x"] = rng.uniform(5, 995, n2); kp2["y"] = rng.uniform(5, 745, n2)
    c.SetFeatures(0, kp1, d1.astype(np.float32), 1000, 750); c.SetFeatures(1, kp2, d2.astype(np.float32), 1000, 750)
    idx, b1, b2 = o.bf_match(d1, d2)
    m, g1, g2 = c.BFMatch(0, 1, sorted_=False)
    ok = len(m) == n1 and np.array_equal(m["trainIdx"], idx) and np.array_equal(g1, b1) and (n2 < 2 or np.array_equal(g2, b2))
    n += 1
    if not ok: bad += 1; print("MISMATCH", n1, n2, kind, int((m["trainIdx"] != idx).sum()), int((g1 != b1).sum()), int((g2 != b2).sum()) if n2 > 1 else 0, flush=True)
print("match soak: %d pairs, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
