import numpy as np, hashlib, time, sys
sys.path.insert(0, "/root/repo")
import imagemosaicing_amd as im
rng = np.random.default_rng(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500; win = 182
pairs = [(i, j) for i in range(N) for j in range(i + 1, min(N, i + win)) if (j == i + 1 or rng.random() < 0.02)]
r = np.zeros(len(pairs), im.PAIR_RESULT)
pos = np.cumsum(rng.uniform(300, 900, (N, 2)), axis=0)
for k, (i, j) in enumerate(pairs):
    n = 150
    xy = rng.uniform(0, 4000, (n, 2)).astype(np.float32)
    r["i"][k] = i; r["j"][k] = j; r["n_in"][k] = n; r["accepted"][k] = 1; r["ok"][k] = 1
    r["a"]["x"][k, :n] = xy[:, 0] + (pos[j, 0] - pos[i, 0]) + rng.normal(0, .3, n); r["a"]["y"][k, :n] = xy[:, 1] + (pos[j, 1] - pos[i, 1]) + rng.normal(0, .3, n)
    r["b"]["x"][k, :n] = xy[:, 0]; r["b"]["y"][k, :n] = xy[:, 1]
T = im.global_affine_align_results(r, N)
t0 = time.perf_counter()
for _ in range(5): T = im.global_affine_align_results(r, N)
dt = (time.perf_counter() - t0) / 5
print(hashlib.sha1(T["m"].tobytes()).hexdigest()[:16], "%.2f ms" % (dt * 1e3), len(pairs), float(np.abs(T["m"][:, 2] - (pos[:, 0] - pos[0, 0])).max()))
