"""Seeded synthetic inputs shared by the golden-vector generator and the tests (so the committed
expected outputs and the test-time inputs are the same bytes)."""
import numpy as np
from tests.oracle_lib import sfpoints


def synth_pairs(n, out_frac, seed, size=(1000, 750), noise=0.3):
    """n correspondences p1 = H(p2) + N(0,noise), the first out_frac*n replaced by uniform outliers."""
    w, h = size
    rng = np.random.default_rng(seed)
    H = np.array([1.01, 0.02, 30, -0.015, 0.99, -80, 1e-5 * 1000 / w, -2e-5 * 1000 / w, 1.0])
    p2 = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1)
    d = H[6] * p2[:, 0] + H[7] * p2[:, 1] + 1
    p1 = np.stack([(H[0] * p2[:, 0] + H[1] * p2[:, 1] + H[2]) / d, (H[3] * p2[:, 0] + H[4] * p2[:, 1] + H[5]) / d], 1)
    p1 = p1 + rng.normal(0, noise, (n, 2))
    no = int(out_frac * n)
    p1[:no] = np.stack([rng.uniform(0, w, no), rng.uniform(0, h, no)], 1)
    return sfpoints(p1), sfpoints(p2)


def texture(w, h, seed):
    """BGR u8 test texture (smooth waves + noise) with IplImage-style contiguous rows."""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.uint8)
    for c in range(3):
        img[..., c] = (127 + 60 * np.sin(xx * 0.05 * (c + 1)) + 50 * np.cos(yy * 0.07 + c) + r.integers(-10, 10, (h, w))).clip(0, 255)
    return img


def warp_cases():
    Hs = [np.eye(3),
          np.array([[1, 0, 10.3], [0, 1, -7.6], [0, 0, 1]]),
          np.array([[0.9, 0.2, 5], [-0.2, 0.95, 3], [0, 0, 1]]),
          np.array([[1.05, 0.03, 12.5], [-0.04, 0.97, -3.25], [2e-5, -3e-5, 1]]),
          np.array([[0.8, -0.1, 0], [0.15, 1.1, 20], [-1e-4, 5e-5, 1.0]]),
          # row 1 of the reference's Release/tran0.txt (affine)
          np.array([[0.993179, -0.00158966, 11.5839], [-0.00284615, 0.988366, -100.622], [0, 0, 1]])]
    return [h.reshape(9).astype(np.float32) for h in Hs]


def mosaic_case():
    imgs = [texture(320, 240, s) for s in range(4)]
    Hs = warp_cases()
    h9s = np.stack([Hs[0], Hs[1], Hs[3], Hs[2]]).astype(np.float32)
    h9s[1, 2] += 150
    h9s[2, 5] += 120
    h9s[3, 2] -= 100
    return imgs, h9s
