"""randomised parity soak of the warps (single image, mosaic canvas, chips+masks+blend): GPU vs oracle"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth import texture
o = oracle_lib.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0
def randH(scale):
    H = np.eye(3) + rng.normal(0, 0.08, (3, 3))
    H[0, 2] = rng.uniform(-scale, scale); H[1, 2] = rng.uniform(-scale, scale)
    H[2, 0] = rng.normal(0, 2e-4); H[2, 1] = rng.normal(0, 2e-4); H[2, 2] = 1
    return H.reshape(9).astype(np.float32)
while time.time() - t0 < budget:
    w = int(rng.integers(40, 700)); h = int(rng.integers(40, 500))
    img = texture(w, h, seed=int(rng.integers(1 << 30)))
    h9 = randH(80)
    rc, ref = o.image_projection_transform(img, h9)
    try:
        buf, dw, dh, dws = c.ImageProjectionTransform(img, h9)
        ok = rc == 0 and (dw, dh, dws) == ref[1:] and np.array_equal(buf, ref[0])
    except Exception as e:
        ok = rc != 0
    n += 1
    if not ok: bad += 1; print("WARP MISMATCH", w, h, h9.tolist(), flush=True)
    # small mosaics: 2-4 images, affine-ish transforms, blend on top
    k = int(rng.integers(2, 5))
    imgs = [texture(int(rng.integers(60, 300)), int(rng.integers(60, 200)), seed=int(rng.integers(1 << 30))) for _ in range(k)]
    h9s = np.stack([randH(150) for _ in range(k)]); h9s[:, 6:8] = 0; h9s[0] = np.eye(3).reshape(9)
    ref_c = o.chips_and_masks(imgs, h9s, find_masks=True)
    got_c = c.ChipsAndMasks(imgs, h9s, find_masks=True)
    ok = len(ref_c["chips"]) == len(got_c["chips"]) and all(np.array_equal(a, b) for a, b in zip(ref_c["chip_imgs"], got_c["chip_imgs"])) and all(np.array_equal(a, b) for a, b in zip(ref_c["masks"], got_c["masks"]))
    if ok:
        rb, _ = o.multiband_blend(ref_c["chips"], ref_c["chip_imgs"], ref_c["masks"], ref_c["cw"], ref_c["ch"], band=5)
        gb, _, _, _ = c.MosaicBlended(imgs, h9s, band=5)
        ok = np.array_equal(rb, gb)
    n += 1
    if not ok: bad += 1; print("MOSAIC MISMATCH", k, [i.shape for i in imgs], h9s.tolist(), flush=True)
print("warp soak: %d cases, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
