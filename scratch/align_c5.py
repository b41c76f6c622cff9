import numpy as np, time, sys, os
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
rng = np.random.default_rng(3)
N, per = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 25
def rc(k):
    r, c = divmod(k, per)
    if r & 1: c = per - 1 - c
    return r, c
pairs = []
for i in range(N):
    ri, ci = rc(i)
    for j in range(i + 1, min(N, i + 182)):
        rj, cj = rc(j)
        if (rj == ri and abs(cj - ci) <= 14) or (rj in (ri + 1, ri + 2, ri + 3) and abs(cj - ci) <= 7):
            if rng.random() < 0.8: pairs.append((i, j))
r = np.zeros(len(pairs), im.PAIR_RESULT)
pos = np.array([[rc(k)[1] * 1600.0, rc(k)[0] * 2100.0] for k in range(N)]) + rng.uniform(-40, 40, (N, 2))
for k, (i, j) in enumerate(pairs):
    n = 237
    xy = rng.uniform(0, 4000, (n, 2)).astype(np.float32)
    r["i"][k] = i; r["j"][k] = j; r["n_in"][k] = n; r["accepted"][k] = 1; r["ok"][k] = 1
    r["a"]["x"][k, :n] = xy[:, 0] + (pos[j, 0] - pos[i, 0]) + rng.normal(0, .3, n); r["a"]["y"][k, :n] = xy[:, 1] + (pos[j, 1] - pos[i, 1]) + rng.normal(0, .3, n)
    r["b"]["x"][k, :n] = xy[:, 0]; r["b"]["y"][k, :n] = xy[:, 1]
T = im.global_affine_align_results(r, N)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); T = im.global_affine_align_results(r, N); ts.append((time.perf_counter() - t0) * 1e3)
import hashlib
print("threads", os.environ.get("MI355_HOST_THREADS"), "pairs", len(pairs), "max |i-j|", max(j - i for i, j in pairs), "align min %.2f ms median %.2f" % (min(ts), sorted(ts)[3]), hashlib.sha1(T["m"].tobytes()).hexdigest()[:12], float(np.abs(T["m"][:, 2] - (pos[:, 0] - pos[0, 0])).max()))
