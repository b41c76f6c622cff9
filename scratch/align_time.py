"""mi355_global_affine_align at C5 size on the host: 2000 images in a block layout (45 per row), pairs up to two rows apart"""
import sys, time, hashlib, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
rng = np.random.default_rng(1)
F, per = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 45
pos = np.stack([(np.arange(F) % per) * 400.0, (np.arange(F) // per) * 300.0], 1)
pairs = [(i, i + d) for i in range(F) for d in (1, per - 1, per, per + 1, 2 * per) if i + d < F and abs(pos[i, 0] - pos[i + d, 0]) < 1500]
K = 40
mp = np.zeros(len(pairs) * K, im.MATCHPAIR)
print(mp.dtype.names)
for p, (a, b) in enumerate(pairs):
    s = slice(p * K, (p + 1) * K)
    xa = rng.uniform(0, 4000, K); ya = rng.uniform(0, 3000, K)
    mp["ai"][s] = a; mp["bi"][s] = b
    mp["ax"][s] = xa; mp["ay"][s] = ya
    mp["bx"][s] = xa + pos[a, 0] - pos[b, 0] + rng.normal(0, 0.3, K); mp["by"][s] = ya + pos[a, 1] - pos[b, 1] + rng.normal(0, 0.3, K)
t0 = time.perf_counter(); T = im.global_affine_align(mp, F); dt = time.perf_counter() - t0
print("pairs", len(pairs), "align %.1f ms" % (dt * 1e3), "sha", hashlib.sha1(T.tobytes()).hexdigest()[:12])
t0 = time.perf_counter(); T = im.global_affine_align(mp, F); dt = time.perf_counter() - t0
print("again %.1f ms" % (dt * 1e3), T["m"][7][:6])
