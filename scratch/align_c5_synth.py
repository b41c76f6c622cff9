"""C5-sized alignment on the host alone (no GPU): synthetic moments of the block layout's overlapping window pairs -> select_connected +
global_affine_align_moments, stage times with MI355_ALIGN_DBG=1.   python scratch/align_c5_synth.py [reps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import imagemosaicing_amd as im
from tests.synth_survey import block_layout, affine3
F, w, h = 2000, 4000, 3000
A = block_layout(F, w, h, seed=5)
M = [affine3(a) for a in A]
c = np.array([(m @ np.array([w / 2, h / 2, 1]))[:2] for m in M])
rng = np.random.default_rng(1)
pairs = [(i, j) for i in range(F) for j in range(i + 1, min(F, i + 182)) if abs(c[i, 0] - c[j, 0]) < 0.8 * w and abs(c[i, 1] - c[j, 1]) < 0.8 * h]
mom = np.zeros(len(pairs), im.PAIR_MOMENTS)
for k, (i, j) in enumerate(pairs):
    n = 60
    pj = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n), np.ones(n)])
    pi = np.linalg.inv(M[i]) @ M[j] @ pj
    ca = np.stack([pi[0], pi[1], np.ones(n)]); cb = pj
    aa = ca @ ca.T; ab = ca @ cb.T; bb = cb @ cb.T
    mom[k]["i"] = i; mom[k]["j"] = j; mom[k]["n_in"] = n
    mom[k]["aa"] = [aa[0, 0], aa[1, 0], aa[1, 1], aa[2, 0], aa[2, 1], aa[2, 2]]
    mom[k]["ab"] = ab.reshape(9)
    mom[k]["bb"] = [bb[0, 0], bb[1, 0], bb[1, 1], bb[2, 0], bb[2, 1], bb[2, 2]]
print(len(pairs), "pairs")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    label = im.select_connected_moments(mom, F); label[0] = 1
    t1 = time.perf_counter()
    fixed = [1 if (k == 0 or label[k] == 0) else 0 for k in range(F)]
    T = im.global_affine_align_moments(mom, F, fixed=fixed, label=label)
    t2 = time.perf_counter()
    print("select %.2f ms align %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
err = max(np.abs(T["m"][k][:6].reshape(2, 3) - (np.linalg.inv(M[0]) @ M[k])[:2]).max() for k in range(F))
print("max abs error of the transforms against the ground truth", err)
