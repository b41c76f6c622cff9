"""randomised parity soak of LaplacianPyramidBlending in one call (mi355_mosaic_blended: validity masks, ownership, owned boxes, active
windows, deferred chip pixels, batched level 0): piles of overlapping frames -- similarity and mildly projective maps, duplicated frames
(chips that own nothing), keep[] subsets, bands 0 .. 6 -- against the oracle's blend of the same chips and masks (which are pinned on
their own by the chips / masks goldens)"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth import texture
o = oracle_lib.load_oracle_fast()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
c = im.Context(0)
t0 = time.time(); n = 0; bad = 0; ns = 0; sbad = 0
while time.time() - t0 < budget:
    k = int(rng.integers(2, 48))
    w = int(rng.integers(48, 460)); h = int(rng.integers(40, 340))
    sizes = [(w, h)] * k if rng.random() < 0.6 else [(int(rng.integers(48, 460)), int(rng.integers(40, 340))) for _ in range(k)]
    imgs = [texture(a, b, seed=int(rng.integers(1 << 30))) for a, b in sizes]
    spread = float(rng.choice([40, 200, 700, 2500])); yawmax = float(rng.choice([0, 3, 25])); proj = float(rng.choice([0, 0, 2e-5]))
    h9s = np.zeros((k, 9), np.float32)
    for q in range(k):
        yaw = np.deg2rad(rng.uniform(-yawmax, yawmax)); s = 1 + rng.uniform(-0.05, 0.05)
        H = np.array([[s * np.cos(yaw), -s * np.sin(yaw), rng.uniform(0, spread)], [s * np.sin(yaw), s * np.cos(yaw), rng.uniform(0, spread)],
                      [rng.normal(0, proj), rng.normal(0, proj), 1.0]])
        h9s[q] = H.reshape(9)
    h9s[0] = np.eye(3).reshape(9)
    for q in range(1, k):
        if rng.random() < 0.12:                      # an exact duplicate of an earlier frame: owns nothing
            p = int(rng.integers(0, q)); h9s[q] = h9s[p]; imgs[q] = imgs[p].copy()
    keep = None if rng.random() < 0.6 else (rng.random(k) < 0.8).astype(np.uint8)
    if keep is not None: keep[0] = 1
    band = int(rng.choice([0, 1, 2, 3, 5, 5, 6]))
    try:
        r = c.ChipsAndMasks(imgs, h9s, keep=keep, find_masks=True)
    except Exception as e:                           # (a map that is not invertible, an empty chip: both paths refuse the same way)
        continue
    if r["cw"] * r["ch"] > 30e6: continue
    ref, _ = o.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=band)
    got, ow, oh, _ = c.MosaicBlended(imgs, h9s, keep=keep, band=band)
    ok = (ow, oh) == (r["cw"], r["ch"]) and np.array_equal(got, ref)
    n += 1
    if not ok: bad += 1; print("BLEND MISMATCH", k, sizes[:3], spread, yawmax, proj, band, None if keep is None else keep.tolist(), flush=True)
    # the same canvas as stripes (mi355_mosaic_blended_rows_dev): a random cut into 2 .. 9 stripes must give the same bytes
    if ok:
        import torch
        d_imgs = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in imgs]
        ptrs = [t.data_ptr() for t in d_imgs]
        wv = [a.shape[1] for a in imgs]; hv = [a.shape[0] for a in imgs]; wsv = [a.strides[0] for a in imgs]
        cuts = sorted(set([0, oh] + [int(v) for v in rng.integers(1, max(oh, 2), int(rng.integers(1, 9)))]))
        sok = True
        for a, b in zip(cuts[:-1], cuts[1:]):
            part, _, _, _ = c.MosaicBlendedDev(ptrs, wv, hv, wsv, h9s, keep=keep, band=band, row0=a, rows=b - a)
            sok = sok and np.array_equal(part.cpu().numpy(), got[a:b])
        ns += 1
        if not sok: sbad += 1; print("STRIPE MISMATCH", k, sizes[:3], spread, yawmax, proj, band, cuts, flush=True)
print("blend soak: %d mosaics, %d mismatches; %d cut into stripes, %d mismatches, %.0f s" % (n, bad, ns, sbad, time.time() - t0))
