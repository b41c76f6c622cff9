"""randomised parity soak of SelectMatchPairs alone: GPU vs oracle; keypoints on cell borders / outside the image, duplicated
indices, tiny and huge grids, nMatch at and around the caps"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib as ol
o = ol.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    K = int(rng.integers(1, 2049)); w = int(rng.integers(8, 4001)); h = int(rng.integers(8, 3001)); kind = int(rng.integers(0, 5))
    gx = int(rng.choice([1, 2, 3, 3, 3, 5])); gy = int(rng.choice([1, 2, 3, 3, 3, 4]))
    k1 = np.stack([rng.uniform(0, w, K), rng.uniform(0, h, K)], 1).astype(np.float32)
    k2 = np.stack([rng.uniform(0, w, K), rng.uniform(0, h, K)], 1).astype(np.float32)
    if kind == 1:      # exactly on the cell borders and on the image border
        k1[:, 0] = np.round(k1[:, 0] / (w / gx)) * np.float32(w / gx); k1[:, 1] = np.round(k1[:, 1] / (h / gy)) * np.float32(h / gy)
    elif kind == 2:    # outside the image (negative, beyond)
        k1[: K // 3] -= np.float32(w); k1[K // 3: 2 * K // 3] += np.float32(w)
    elif kind == 3:    # integer coordinates, many duplicates
        k1 = np.round(k1 / 97) * 97; k2 = np.round(k2 / 97) * 97
    M = int(rng.integers(0, K + 1)) if rng.random() < 0.3 else K
    m = np.stack([rng.permutation(K)[:M] if kind != 4 else rng.integers(0, K, M), rng.integers(0, K, M)], 1).astype(np.int32)
    nm = min(400, int(rng.choice([0, 1, 4, 99, 396, 400, M, max(M - 1, 0)])))
    a1, a2 = c.SelectMatchPairs(m, k1, k2, nm, w, h, gx, gy) if M else (np.zeros(0, ol.SFPOINT),) * 2
    b1, b2 = o.select(m, k1, k2, nm, w, h, gx, gy) if M else (np.zeros(0, ol.SFPOINT),) * 2
    good = len(a1) == len(b1) and np.array_equal(a1.view(np.uint8), b1.view(np.uint8)) and np.array_equal(a2.view(np.uint8), b2.view(np.uint8))
    n += 1
    if not good: bad += 1; print("MISMATCH kind", kind, "K", K, "M", M, "nm", nm, w, h, gx, gy, len(a1), len(b1), flush=True)
print("select soak: %d cases, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
