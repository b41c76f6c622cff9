"""hashes of mi355_global_affine_align_results over surveys with several fixed-image sets and labels: the refactored host path (moments first,
right-hand sides derived from them) and the moments-from-the-GPU path must give the same bits"""
import numpy as np, hashlib, sys
sys.path.insert(0, "/root/repo")
import imagemosaicing_amd as im
out = []
for seed, N in ((3, 120), (4, 500), (5, 37)):
    rng = np.random.default_rng(seed); win = 182
    pairs = [(i, j) for i in range(N) for j in range(i + 1, min(N, i + win)) if (j == i + 1 or rng.random() < 0.03)]
    if seed == 5: pairs += [(9, 3), (20, 20)]                      # i > j, i == j
    r = np.zeros(len(pairs), im.PAIR_RESULT)
    pos = np.cumsum(rng.uniform(300, 900, (N, 2)), axis=0)
    for k, (i, j) in enumerate(pairs):
        n = int(rng.integers(31, 400))
        xy = rng.uniform(0, 4000, (n, 2)).astype(np.float32)
        r["i"][k] = i; r["j"][k] = j; r["n_in"][k] = n; r["accepted"][k] = 1; r["ok"][k] = 1
        r["a"]["x"][k, :n] = xy[:, 0] + (pos[j, 0] - pos[i, 0]) + rng.normal(0, .3, n); r["a"]["y"][k, :n] = xy[:, 1] + (pos[j, 1] - pos[i, 1]) + rng.normal(0, .3, n)
        r["b"]["x"][k, :n] = xy[:, 0]; r["b"]["y"][k, :n] = xy[:, 1]
    r["accepted"][::17] = 0
    fixed_sets = [None, [1 if k in (0,) else 0 for k in range(N)], [1 if k % 9 == 4 else 0 for k in range(N)], [1 if k >= N - 3 else 0 for k in range(N)]]
    for fs in fixed_sets:
        for lab in (None, np.array([0 if k % 11 == 5 else 1 for k in range(N)], np.int32)):
            T = im.global_affine_align_results(r, N, fixed=fs, label=lab)
            out.append(hashlib.sha1(T.tobytes()).hexdigest()[:12])
print(" ".join(out))
