"""CPU: pieces of the blend path that are now pinned by the reference's own code (oracle/_ref, VERDICT r01 "next" #7 / #8):

  * mi355_resample_by_overlap (product, host C++) against the reference's ResampleByOverlap (MosaicImage.cpp:2069-2201),
    on random layouts where oracle/_ref exists and on the committed vectors everywhere;
  * the oracle's chips + distance-map masks (oracle_warp.c) against the reference's LaplacianPyramidBlending warp stage
    (:2216-2460) + FindMasksByDistMap (:1761-1881), same two ways."""
import numpy as np

from tests.golden_util import blend_golden, bits
from tests.synth import texture, mosaic_case


def overlap_layouts(seed, n_cases=12):
    """strips with varying forward overlap (some images nearly on top of each other), yaw, mild projective terms, skipped images"""
    rng = np.random.default_rng(seed)
    out = []
    for c in range(n_cases):
        n = int(rng.integers(3, 14))
        w, h = [(640, 480), (1000, 750), (320, 240)][c % 3]
        h9 = np.zeros((n, 9), np.float32)
        x = 0.0
        for k in range(n):
            x += w * rng.choice([0.05, 0.15, 0.25, 0.4, 0.8])
            a = np.deg2rad(rng.uniform(-8, 8)); s = 1 + rng.uniform(-0.05, 0.05)
            H = np.array([[s * np.cos(a), -s * np.sin(a), x + rng.uniform(-5, 5)], [s * np.sin(a), s * np.cos(a), rng.uniform(-0.2, 0.2) * h],
                          [rng.normal(0, 2e-5) if c % 2 else 0.0, rng.normal(0, 2e-5) if c % 2 else 0.0, 1.0]])
            h9[k] = H.reshape(9)
        if n > 5 and c % 4 == 0:
            h9[2, 8] = 0.0
        out.append(([w] * n, [h] * n, h9))
    return out


def test_resample_by_overlap_vs_reference(lib, ref):
    total, dropped = 0, 0
    for seed in (1, 2, 3):
        for w, h, h9 in overlap_layouts(seed):
            got = lib.resample_by_overlap(w, h, h9, 0.7)
            want = ref.resample_by_overlap(w, h, h9, 0.7)
            assert np.array_equal(got, want), (seed, got, want)
            total += len(w); dropped += int((want == 0).sum())
    assert 0 < dropped < total // 2


def test_resample_by_overlap_golden(lib):
    g = blend_golden()
    for k, (w, h, h9) in enumerate(overlap_layouts(7)):
        assert np.array_equal(lib.resample_by_overlap(w, h, h9, 0.7), g[f"keep{k}"]), k
    assert sum(int((g[f"keep{k}"] == 0).sum()) for k in range(12)) > 0


def blend_cases():
    imgs, h9s = mosaic_case()
    yield "mosaic", imgs, h9s
    rng = np.random.default_rng(12)
    imgs2 = [texture(200, 150, seed=30 + k) for k in range(6)]
    h2 = np.zeros((6, 9), np.float32)
    for k in range(6):
        a = np.deg2rad(rng.uniform(-10, 10))
        H = np.array([[np.cos(a), -np.sin(a), 70.0 * k + 20], [np.sin(a), np.cos(a), 15.0 * (k % 3) + 30], [1e-5 * k, -2e-5, 1.0]])
        h2[k] = H.reshape(9)
    h2[4, 8] = 0
    yield "strip", imgs2, h2


def _same(a, b, tag):
    assert (a["cw"], a["ch"]) == (b["cw"], b["ch"]) and len(a["chips"]) == len(b["chips"]), tag
    for k in range(len(b["chips"])):
        for f in ("x0", "y0", "w", "h", "img"):
            assert int(a["chips"][k][f]) == int(b["chips"][k][f]), (tag, k, f)
        assert np.array_equal(bits(a["chips"][k]["quad"]), bits(b["chips"][k]["quad"])), (tag, k)
        assert np.array_equal(a["chip_imgs"][k], b["chip_imgs"][k]), (tag, k, "chip")
        w = int(b["chips"][k]["w"])            # row padding is not image data (the reference's memset(255) covers it, cvZero clears it)
        assert np.array_equal(a["masks"][k][:, :w], b["masks"][k][:, :w]), (tag, k, "mask")


def test_oracle_chips_and_masks_vs_reference(oracle, ref):
    for tag, imgs, h9s in blend_cases():
        for fm in (True, False):
            a = oracle.chips_and_masks(imgs, h9s, find_masks=fm)
            b = ref.chips_and_masks(imgs, h9s, keep=np.ones(len(imgs), np.uint8), find_masks=fm)
            _same(a, b, f"{tag} find_masks={fm}")
    # with the reference's own ResampleByOverlap deciding what is kept
    imgs, h9s = list(blend_cases())[1][1:]
    keep = ref.resample_by_overlap([i.shape[1] for i in imgs], [i.shape[0] for i in imgs], h9s, 0.7)
    _same(oracle.chips_and_masks(imgs, h9s, keep=keep), ref.chips_and_masks(imgs, h9s, keep=None), "strip resampled")


def test_oracle_chips_and_masks_golden(oracle):
    g = blend_golden()
    for tag, imgs, h9s in blend_cases():
        a = oracle.chips_and_masks(imgs, h9s, find_masks=True)
        assert [a["cw"], a["ch"], len(a["chips"])] == g[f"{tag}_dims"].tolist()
        for k in range(len(a["chips"])):
            assert np.array_equal(a["chip_imgs"][k], g[f"{tag}_chip{k}"]) and np.array_equal(a["masks"][k], g[f"{tag}_mask{k}"])
            assert np.array_equal(bits(a["chips"][k]["quad"]), g[f"{tag}_quad{k}"])
