"""Seeded synthetic "UAV terrain" frames for tests and the CPU side of the bench (numpy / scipy only).

terrain(): sum of value-noise octaves + anisotropic Gaussian blobs + line segments, BGR u8.
frames are cut from it through ground-truth similarity transforms with forward overlap, yaw / scale jitter,
per-frame gain and additive noise."""
import numpy as np
from scipy import ndimage


def terrain(w, h, seed=0):
    rng = np.random.default_rng(seed)
    acc = np.zeros((h, w), np.float32)
    amp = 1.0
    for o in range(7):
        cell = max(3, 192 >> o)
        gh, gw = h // cell + 3, w // cell + 3
        g = rng.random((gh, gw)).astype(np.float32)
        up = ndimage.zoom(g, cell, order=1)[:h, :w]
        if up.shape != (h, w):
            up = np.pad(up, ((0, h - up.shape[0]), (0, w - up.shape[1])), mode="edge")
        acc += amp * up
        amp *= 0.7
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    img = np.stack([acc * 150 + 40, acc * 170 + 30, acc * 120 + 60], -1)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nblob = max(40, (w * h) // 250)
    for _ in range(nblob):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sx, sy = rng.uniform(1.2, 6), rng.uniform(1.2, 6)
        x0, x1 = int(max(0, cx - 4 * sx)), int(min(w, cx + 4 * sx) + 1)
        y0, y1 = int(max(0, cy - 4 * sy)), int(min(h, cy + 4 * sy) + 1)
        if x1 <= x0 or y1 <= y0:
            continue
        gx = np.exp(-0.5 * ((xx[y0:y1, x0:x1] - cx) / sx) ** 2 - 0.5 * ((yy[y0:y1, x0:x1] - cy) / sy) ** 2)
        col = rng.uniform(-90, 90, 3)
        img[y0:y1, x0:x1] += gx[..., None] * col
    for _ in range(max(5, (w * h) // 6000)):
        x0, y0 = rng.uniform(0, w), rng.uniform(0, h)
        ang, ln = rng.uniform(0, np.pi), rng.uniform(30, 200)
        n = int(ln)
        xs = (x0 + np.cos(ang) * np.arange(n)).astype(int)
        ys = (y0 + np.sin(ang) * np.arange(n)).astype(int)
        ok = (xs >= 1) & (xs < w - 1) & (ys >= 1) & (ys < h - 1)
        col = rng.uniform(-80, 80, 3)
        for dy in (0, 1):
            img[ys[ok] + dy, xs[ok]] += col
    return np.clip(img, 0, 255).astype(np.uint8)


def cut_frame(tex, w, h, cx, cy, yaw_deg=0.0, scale=1.0, gain=1.0, noise=0.0, seed=0):
    """frame pixel (x,y) <- tex at  c + s R (x - w/2, y - h/2); returns (BGR u8 frame, 3x3 H frame->tex)"""
    a = np.deg2rad(yaw_deg)
    R = scale * np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    t = np.array([cx, cy]) - R @ np.array([w / 2.0, h / 2.0])
    H = np.eye(3)
    H[:2, :2] = R
    H[:2, 2] = t
    # ndimage works in (row, col): matrix maps output coords -> input coords
    M = np.array([[R[1, 1], R[1, 0]], [R[0, 1], R[0, 0]]])
    off = np.array([t[1], t[0]])
    out = np.stack([ndimage.affine_transform(tex[..., c].astype(np.float32), M, offset=off, output_shape=(h, w), order=1, mode="reflect")
                    for c in range(3)], -1)
    rng = np.random.default_rng(seed)
    out = out * gain + (rng.normal(0, noise, out.shape) if noise > 0 else 0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), H


def strip(n, w, h, seed=0, overlap=0.6):
    """n frames along a strip with forward overlap; returns frames and ground-truth H (frame k -> texture)"""
    step = int(w * (1 - overlap))
    tw, th = w + step * (n - 1) + 200, h + 200
    tex = terrain(tw, th, seed=seed)
    rng = np.random.default_rng(seed + 1)
    frames, Hs = [], []
    for k in range(n):
        f, H = cut_frame(tex, w, h, 100 + w / 2 + k * step, 100 + h / 2 + rng.uniform(-8, 8), yaw_deg=rng.uniform(-3, 3),
                         scale=1 + rng.uniform(-0.02, 0.02), gain=1 + rng.uniform(-0.05, 0.05), noise=2.0, seed=seed * 1000003 + k)
        frames.append(f)
        Hs.append(H)
    return frames, Hs


def two_tiles(w=640, h=480, seed=1):
    frames, _ = strip(2, w, h, seed=seed)
    return frames[0], frames[1]
