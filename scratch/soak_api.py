"""state-machine soak of the detect+describe entry points: random interleavings of device / host extraction (two frame sizes),
re-extraction of an id that is still in flight, GetFeatures / DropFeatures / MatchPairs on subsets, option changes between
contexts -- every feature set that comes back must be the oracle's for the content that was submitted LAST under that id"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth_frames import terrain
o = oracle_lib.load_oracle_fast()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
sizes = [(640, 480), (1000, 750), (333, 257), (2048, 1536)]
pool = []                                     # (img, oracle kp, oracle desc)
for (w, h) in sizes:
    for s in range(3):
        img = terrain(w, h, seed=100 * w + s)
        kp, d = o.sift(img)
        pool.append((img, kp, d))
t0 = time.time(); checks = 0; bad = 0; ops = 0
while time.time() - t0 < budget:
    ctxs = []
    for _ in range(2):                        # two contexts alive at once, operations alternate between them at random
        c = im.Context(0)
        c.set_option("sift_batch", int(rng.integers(1, 9))); c.set_option("sift_slots", int(rng.integers(1, 5)))
        c.set_option("sift_cascade", int(rng.integers(0, 4)))
        ctxs.append((c, {}))
    keep = []                                 # device tensors must stay alive until their batch ran
    for step in range(int(rng.integers(5, 80))):
        ops += 1
        c, cur = ctxs[int(rng.integers(0, 2))]   # cur: id -> pool index
        op = rng.random()
        if op < 0.55:
            k = int(rng.integers(0, 12)); pi = int(rng.integers(0, len(pool))); img = pool[pi][0]
            h, w = img.shape[:2]
            if rng.random() < 0.75:
                d = torch.from_numpy(np.ascontiguousarray(img)).cuda(); torch.cuda.synchronize(); keep.append(d)
                c.SiftExtractDev(k, d.data_ptr(), w, h, w * 3)
            else:
                (c.SiftExtractHost(k, img) if rng.random() < 0.5 else c.SiftExtract(k, img))
            cur[k] = pi
        elif op < 0.75 and cur:
            k = int(rng.choice(list(cur)))
            kp, desc = c.GetFeatures(k)
            _, okp, od = pool[cur[k]]
            ok = len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) and np.array_equal(desc.astype(np.uint8), od)
            checks += 1
            if not ok: bad += 1; print("MISMATCH get", k, cur[k], len(kp), len(okp), flush=True)
        elif op < 0.85 and cur:
            k = int(rng.choice(list(cur))); c.DropFeatures(k); del cur[k]
        elif op < 0.95 and len(cur) >= 2:
            ids = list(cur); i, j = (int(v) for v in rng.choice(ids, 2, replace=False))
            r = c.MatchPairs([(i, j)], 2.5, 3)[0]
            (_, k1, d1), (_, k2, d2) = pool[cur[i]], pool[cur[j]]
            hh, ww = pool[cur[i]][0].shape[:2]
            nin, i1, i2, H, ns = o.match_pair(np.stack([k1["x"], k1["y"]], 1), d1, np.stack([k2["x"], k2["y"]], 1), d2, ww, hh, 2.5, 3)
            ok = int(r["n_selected"]) == ns and int(r["accepted"]) == int(nin > 30) and (nin <= 30 or (int(r["n_in"]) == nin and np.array_equal(r["H"].view(np.uint32), H.view(np.uint32))))
            checks += 1
            if not ok: bad += 1; print("MISMATCH pair", i, j, cur[i], cur[j], int(r["n_selected"]), ns, int(r["n_in"]), nin, flush=True)
        else:
            c.synchronize()
    for c, cur in ctxs:
        for k, pi in cur.items():
            kp, desc = c.GetFeatures(k)
            _, okp, od = pool[pi]
            ok = len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) and np.array_equal(desc.astype(np.uint8), od)
            checks += 1
            if not ok: bad += 1; print("MISMATCH final", k, pi, len(kp), len(okp), flush=True)
    for c, _ in ctxs: c.close()
    keep.clear()
print("api soak: %d operations, %d checks, %d mismatches, %.0f s" % (ops, checks, bad, time.time() - t0))
