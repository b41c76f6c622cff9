"""Synthetic UAV survey layout shared by bench.py and the config-sized GPU tests: a serpentine strip of frames
over one procedural terrain, 60 % forward / 30 % side overlap, +-3 deg yaw, +-2 % scale, +-5 % gain.  The frames
themselves are rendered straight into HBM by mi355_synth_frame_dev (csrc/synth.hip: a pure function of the ground
coordinate, so overlapping frames see the same ground); this module only holds the ground-truth geometry."""
import numpy as np


def frame_layout(n, w, h, rank=0, seed=0xC0FFEE, per_row=25):
    """returns (A [n,6] frame->ground affine maps, gains [n])"""
    rng = np.random.default_rng(seed + 7919 * rank)
    sx, sy = 0.4 * w, 0.7 * h
    A, gains = [], []
    for k in range(n):
        row, col = divmod(k, per_row)
        if row & 1:
            col = per_row - 1 - col
        cx = w / 2 + col * sx + rng.uniform(-0.01, 0.01) * w
        cy = h / 2 + row * sy + rng.uniform(-0.01, 0.01) * h
        yaw = np.deg2rad(rng.uniform(-3, 3))
        s = 1 + rng.uniform(-0.02, 0.02)
        R = s * np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        t = np.array([cx, cy]) - R @ np.array([w / 2.0, h / 2.0])
        A.append([R[0, 0], R[0, 1], t[0], R[1, 0], R[1, 1], t[1]])
        gains.append(1 + rng.uniform(-0.05, 0.05))
    return np.array(A, np.float64), np.array(gains)


def block_layout(n, w, h, cols=50, seed=5, extent=20000.0):
    """frame -> ground affine maps of a dense block survey whose bounding box is ~20000 x 20000 (C5), +-3 deg yaw,
    +-2 % scale; returns A [n, 6]"""
    rng = np.random.default_rng(seed)
    rows = (n + cols - 1) // cols
    sx = (extent - w) / (cols - 1)
    sy = (extent - h) / (rows - 1)
    A = []
    for k in range(n):
        r, c = divmod(k, cols)
        if r & 1:
            c = cols - 1 - c
        cx, cy = w / 2 + c * sx + rng.uniform(-20, 20), h / 2 + r * sy + rng.uniform(-20, 20)
        yaw = np.deg2rad(rng.uniform(-3, 3)); s = 1 + rng.uniform(-0.02, 0.02)
        R = s * np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        t = np.array([cx, cy]) - R @ np.array([w / 2.0, h / 2.0])
        A.append([R[0, 0], R[0, 1], t[0], R[1, 0], R[1, 1], t[1]])
    return np.array(A, np.float64)



def affine3(a6):
    return np.array([[a6[0], a6[1], a6[2]], [a6[3], a6[4], a6[5]], [0, 0, 1.0]])


def ground_truth_h(A, i, j):
    """homography mapping frame-j pixels onto frame-i pixels (the convention of Ransac2D's H, matrix.h:782)"""
    return np.linalg.inv(affine3(A[i])) @ affine3(A[j])


def render_frames(ctx, torch, n, w, h, rank=0, per_row=25, terrain_seed=None, dev="cuda"):
    """renders the n frames of the layout into one uint8 tensor [n, h*ws] in HBM; returns (frames, A, gains, ws)"""
    ws = (3 * w + 3) & ~3
    A, gains = frame_layout(n, w, h, rank, per_row=per_row)
    frames = torch.empty((n, h * ws), dtype=torch.uint8, device=dev)
    tseed = (0xC0FFEE + 977 * rank) & 0xffffffff if terrain_seed is None else terrain_seed
    for k in range(n):
        ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], tseed, (rank * 1000003 + k) & 0xffffffff, gains[k], 2.0)
    ctx.synchronize()
    return frames, A, gains, ws


def host_image(frames, k, w, h, ws):
    """frame k as a contiguous h x w x 3 numpy array (row padding dropped)"""
    return np.ascontiguousarray(frames[k].cpu().numpy().reshape(h, ws)[:, :3 * w].reshape(h, w, 3))
