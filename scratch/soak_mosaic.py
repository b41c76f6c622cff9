"""randomised parity soak of the one-launch canvas (mosaic_tile_kernel): many overlapping images, projective maps, invalid images,
random stripes -- GPU vs the oracle's image-after-image overwrite"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth import texture
o = oracle_lib.load_oracle_fast()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    k = int(rng.integers(2, 40))
    w = int(rng.integers(40, 420)); h = int(rng.integers(40, 300))
    sizes = [(w, h)] * k if rng.random() < 0.5 else [(int(rng.integers(40, 420)), int(rng.integers(40, 300))) for _ in range(k)]
    imgs = [texture(a, b, seed=int(rng.integers(1 << 30))) for a, b in sizes]
    spread = float(rng.choice([60, 300, 1200])); sc = float(rng.choice([0.02, 0.1, 0.4])); proj = float(rng.choice([0, 1e-4, 6e-4]))
    h9s = np.zeros((k, 9), np.float32)
    for q in range(k):
        H = np.eye(3) + rng.normal(0, sc, (3, 3))
        H[0, 2] = rng.uniform(-spread, spread); H[1, 2] = rng.uniform(-spread, spread)
        H[2, 0] = rng.normal(0, proj); H[2, 1] = rng.normal(0, proj); H[2, 2] = 1
        h9s[q] = H.reshape(9)
    h9s[0] = np.eye(3).reshape(9)
    for q in range(1, k):
        if rng.random() < 0.1: h9s[q, 8] = 0
    rc, ref = o.mosaic_images_refined(imgs, h9s)
    if rc != 0: continue
    ref, rw, rh, rws = ref
    if rws * rh > 400e6: continue
    cw, ch, cws, _ = im.mosaic_layout([i.shape[1] for i in imgs], [i.shape[0] for i in imgs], h9s)
    d_imgs = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    canvas = torch.full((ch, cws), 123, dtype=torch.uint8, device="cuda")
    cuts = sorted(set([0, ch] + [int(x) for x in rng.integers(1, max(ch, 2), int(rng.integers(0, 5)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.MosaicImagesRefinedDev([t.data_ptr() for t in d_imgs], [i.shape[1] for i in imgs], [i.shape[0] for i in imgs], [i.strides[0] for i in imgs], h9s, canvas.data_ptr(), cw, ch, cws, a, b - a)
    c.synchronize()
    ok = (cw, ch, cws) == (rw, rh, rws) and np.array_equal(canvas.cpu().numpy(), ref)
    n += 1
    if not ok: bad += 1; print("MOSAIC MISMATCH", k, sizes[:3], spread, sc, proj, flush=True)
print("mosaic soak: %d canvases, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
