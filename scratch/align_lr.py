import numpy as np, time, sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemosaicing_amd as im
rng = np.random.default_rng(3)
N, per = 500, 25
def rc(k):
    r, c = divmod(k, per)
    if r & 1: c = per - 1 - c
    return r, c
pairs = []
for i in range(N):
    ri, ci = rc(i)
    for j in range(i + 1, min(N, i + 182)):
        rj, cj = rc(j)
        if ((rj == ri and abs(cj - ci) <= 2) or (rj == ri + 1 and abs(cj - ci) <= 2)) and rng.random() < 0.8: pairs.append((i, j))
        elif rng.random() < 0.0004: pairs.append((i, j))          # a few spurious long-range pairs
r = np.zeros(len(pairs), im.PAIR_RESULT)
pos = np.array([[rc(k)[1] * 1600.0, rc(k)[0] * 2100.0] for k in range(N)]) + rng.uniform(-40, 40, (N, 2))
for k, (i, j) in enumerate(pairs):
    n = 60
    xy = rng.uniform(0, 4000, (n, 2)).astype(np.float32)
    r["i"][k] = i; r["j"][k] = j; r["n_in"][k] = n; r["accepted"][k] = 1; r["ok"][k] = 1
    r["a"]["x"][k, :n] = xy[:, 0] + (pos[j, 0] - pos[i, 0]) + rng.normal(0, .3, n); r["a"]["y"][k, :n] = xy[:, 1] + (pos[j, 1] - pos[i, 1]) + rng.normal(0, .3, n)
    r["b"]["x"][k, :n] = xy[:, 0]; r["b"]["y"][k, :n] = xy[:, 1]
T = im.global_affine_align_results(r, N)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); T = im.global_affine_align_results(r, N); ts.append((time.perf_counter() - t0) * 1e3)
# dense normal equations in numpy (double)
D = 3 * (N - 1)
A = np.zeros((D, D)); bx = np.zeros(D); by = np.zeros(D)
for k, (i, j) in enumerate(pairs):
    n = int(r["n_in"][k])
    ca = np.stack([r["a"]["x"][k, :n], r["a"]["y"][k, :n], np.ones(n)], 1).astype(np.float64)
    cb = np.stack([r["b"]["x"][k, :n], r["b"]["y"][k, :n], np.ones(n)], 1).astype(np.float64)
    oa, ob = i - 1, j - 1
    rx = np.zeros(n); ry = np.zeros(n)
    if oa < 0: rx -= ca[:, 0]; ry -= ca[:, 1]
    if oa >= 0: A[3*oa:3*oa+3, 3*oa:3*oa+3] += ca.T @ ca; bx[3*oa:3*oa+3] += ca.T @ rx; by[3*oa:3*oa+3] += ca.T @ ry
    A[3*ob:3*ob+3, 3*ob:3*ob+3] += cb.T @ cb; bx[3*ob:3*ob+3] -= cb.T @ rx; by[3*ob:3*ob+3] -= cb.T @ ry
    if oa >= 0: A[3*oa:3*oa+3, 3*ob:3*ob+3] -= ca.T @ cb; A[3*ob:3*ob+3, 3*oa:3*oa+3] -= cb.T @ ca
X = np.linalg.solve(A, bx); Y = np.linalg.solve(A, by)
err = max(np.abs(T["m"][1:, 0:3].reshape(-1) - X).max(), np.abs(T["m"][1:, 3:6].reshape(-1) - Y).max())
print("threads", os.environ.get("MI355_HOST_THREADS"), "pairs", len(pairs), "max |i-j|", max(j - i for i, j in pairs), "align min %.2f ms" % min(ts), hashlib.sha1(T["m"].tobytes()).hexdigest()[:12], "max abs diff vs numpy dense solve %.3g" % err)
