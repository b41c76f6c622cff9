#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libref_oracle.so, built by
oracle/ref/build_ref.sh from /root/reference) and copies the reference's committed run artefacts
(data files, not source): Release/feature_temp/matchPairs.{match,txt} and Release/tran0.txt.

blend_golden.npz: ResampleByOverlap keep flags and the chips / masks of LaplacianPyramidBlending's warp stage + FindMasksByDistMap.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
Every array pair below is (seeded input, output of the reference's own code compiled with
g++ -O2 -ffp-contract=off, x86-64 SSE2).  The tests compare the oracle restatement and the HIP
path against these bytes.
"""
import os
import shutil
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol          # noqa: E402
from tests.synth import synth_pairs, texture, warp_cases, mosaic_case   # noqa: E402

REL = "/root/reference/code/MosaicingCode/Release"


def main():
    R = ol.load_ref()
    assert R is not None, "needs oracle/_ref (i.e. /root/reference)"
    # ---- data files of the reference's one committed run -------------------------------
    for src, dst in [(f"{REL}/feature_temp/matchPairs.match", "matchPairs.match"),
                     (f"{REL}/feature_temp/matchPairs.txt", "matchPairs.txt"),
                     (f"{REL}/tran0.txt", "tran0.txt")]:
        shutil.copyfile(src, os.path.join(HERE, dst))

    out = {}
    # ---- InverseMatrix -------------------------------------------------------------------
    rng = np.random.default_rng(2024)
    inv_in, inv_eps, inv_rc, inv_out = [], [], [], []
    for order in [2, 3, 8, 13]:
        for eps in [1e-6, 1e-12, 1e-20]:
            for t in range(6):
                a = (rng.normal(size=(order, order)) * rng.choice([1, 100, 1e4])).astype(np.float32)
                if t == 5:
                    a[:, 1] = 0            # singular: no pivot in column 1
                rc, o = R.inverse_matrix(a, eps)
                inv_in.append(np.pad(a, ((0, 13 - order), (0, 13 - order))))
                inv_out.append(np.pad(o, ((0, 13 - order), (0, 13 - order))))
                inv_eps.append(eps)
                inv_rc.append((order, rc))
    out.update(inv_in=np.array(inv_in), inv_out=np.array(inv_out), inv_eps=np.array(inv_eps, np.float32),
               inv_rc=np.array(inv_rc, np.int32))
    # ---- SolveHomographyMatrix + NLLS ----------------------------------------------------
    hp1, hp2, hn, hH, hN = [], [], [], [], []
    for n in [4, 4, 4, 4, 8, 50, 200]:
        for s in range(4):
            p1, p2 = synth_pairs(n, 0.0, seed=1000 * n + s, size=(4000, 3000) if s % 2 else (1000, 750))
            rc, H = R.solve_homography(p1, p2)
            rc, N = R.nlls(p1, p2, H)
            pad = 200 - n
            hp1.append(np.pad(p1, (0, pad)))
            hp2.append(np.pad(p2, (0, pad)))
            hn.append(n)
            hH.append(H)
            hN.append(N)
    out.update(h_p1=np.array(hp1), h_p2=np.array(hp2), h_n=np.array(hn, np.int32), h_H=np.array(hH), h_N=np.array(hN))
    # ---- Ransac2D: synthetic -------------------------------------------------------------
    r_p1, r_p2, r_n, r_seed, r_ok, r_nin, r_ids, r_H = [], [], [], [], [], [], [], []
    cases = [(50, 0.3), (200, 0.2), (396, 0.5), (396, 0.0), (396, 0.7), (30, 0.6), (5, 0.0), (4, 0.0), (3, 0.0), (120, 1.0)]
    for ci, (n, of) in enumerate(cases):
        for seed in (1, 2, 3):
            p1, p2 = synth_pairs(n, of, seed=77 * ci + seed, size=(4000, 3000) if ci % 2 else (1000, 750))
            ok, i1, i2, H = R.ransac2d(p1, p2, 2.5, 1000, seed)
            pad = 400 - n
            r_p1.append(np.pad(p1, (0, pad)))
            r_p2.append(np.pad(p2, (0, pad)))
            r_n.append(n)
            r_seed.append(seed)
            r_ok.append(ok)
            r_nin.append(len(i1))
            r_ids.append(np.pad(i1["id"], (0, 400 - len(i1)), constant_values=-1))
            r_H.append(H)
    # ---- Ransac2D: the reference's own committed correspondences + injected outliers -------
    raw = np.fromfile(os.path.join(HERE, "matchPairs.match"), np.uint8)
    n_rec = int(raw[:4].view(np.int32)[0])
    rec = raw[4:].view(np.dtype([("ax", "<f4"), ("ay", "<f4"), ("aid", "<i4"), ("ai", "<i4"), ("af", "<i4"),
                                 ("bx", "<f4"), ("by", "<f4"), ("bid", "<i4"), ("bi", "<i4"), ("bf", "<i4")]))
    assert len(rec) == n_rec
    pairs = sorted(set(zip(rec["ai"].tolist(), rec["bi"].tolist())))
    rng = np.random.default_rng(5)
    for k, (i, j) in enumerate(pairs):            # every image pair of the reference's committed run
        sel = rec[(rec["ai"] == i) & (rec["bi"] == j)]
        n_out = 100 if k % 2 == 0 else 30
        p1 = np.zeros(len(sel) + n_out, ol.SFPOINT)
        p2 = np.zeros(len(sel) + n_out, ol.SFPOINT)
        p1["x"][:len(sel)] = sel["ax"]; p1["y"][:len(sel)] = sel["ay"]
        p2["x"][:len(sel)] = sel["bx"]; p2["y"][:len(sel)] = sel["by"]
        p1["x"][len(sel):] = rng.uniform(0, 1000, n_out).astype(np.float32)
        p1["y"][len(sel):] = rng.uniform(0, 750, n_out).astype(np.float32)
        p2["x"][len(sel):] = rng.uniform(0, 1000, n_out).astype(np.float32)
        p2["y"][len(sel):] = rng.uniform(0, 750, n_out).astype(np.float32)
        perm = rng.permutation(len(p1))
        p1, p2 = p1[perm], p2[perm]
        p1["id"] = np.arange(len(p1)); p2["id"] = np.arange(len(p1))
        n = len(p1)
        if n > 400:
            p1, p2, n = p1[:400], p2[:400], 400
        seed = 12345 + k
        ok, i1, i2, H = R.ransac2d(p1, p2, 2.5, 1000, seed)
        r_p1.append(np.pad(p1, (0, 400 - n))); r_p2.append(np.pad(p2, (0, 400 - n)))
        r_n.append(n); r_seed.append(seed); r_ok.append(ok); r_nin.append(len(i1))
        r_ids.append(np.pad(i1["id"], (0, 400 - len(i1)), constant_values=-1)); r_H.append(H)
    out.update(r_p1=np.array(r_p1), r_p2=np.array(r_p2), r_n=np.array(r_n, np.int32), r_seed=np.array(r_seed, np.uint32),
               r_ok=np.array(r_ok, np.int32), r_nin=np.array(r_nin, np.int32), r_ids=np.array(r_ids, np.int32), r_H=np.array(r_H))
    # ---- SelectMatchPairs ------------------------------------------------------------------
    s_kp1, s_kp2, s_m, s_wh, s_nm, s_o1, s_o2, s_no = [], [], [], [], [], [], [], []
    rng = np.random.default_rng(9)
    for (w, h, K) in [(640, 480, 2000), (1000, 750, 2000), (1920, 1080, 2000), (4000, 3000, 2000), (4000, 3000, 700), (640, 480, 40)]:
        kp1 = np.stack([rng.uniform(5, w - 5, K), rng.uniform(5, h - 5, K)], 1).astype(np.float32)
        kp2 = np.stack([rng.uniform(5, w - 5, K), rng.uniform(5, h - 5, K)], 1).astype(np.float32)
        m = np.stack([rng.permutation(K), rng.integers(0, K, K)], 1).astype(np.int32)
        nMatch = int(min(400, 0.3 * K))
        o1, o2 = R.select(m, kp1, kp2, nMatch, w, h)
        pad = 2000 - K
        s_kp1.append(np.pad(kp1, ((0, pad), (0, 0)))); s_kp2.append(np.pad(kp2, ((0, pad), (0, 0))))
        s_m.append(np.pad(m, ((0, pad), (0, 0)))); s_wh.append((w, h, K)); s_nm.append(nMatch)
        s_o1.append(np.pad(o1, (0, 400 - len(o1)))); s_o2.append(np.pad(o2, (0, 400 - len(o2)))); s_no.append(len(o1))
    out.update(s_kp1=np.array(s_kp1), s_kp2=np.array(s_kp2), s_m=np.array(s_m), s_wh=np.array(s_wh, np.int32),
               s_nm=np.array(s_nm, np.int32), s_o1=np.array(s_o1), s_o2=np.array(s_o2), s_no=np.array(s_no, np.int32))
    np.savez_compressed(os.path.join(HERE, "math_golden.npz"), **out)

    # ---- warps: ImageProjectionTransform under 6 homographies; 3-image MosaicImagesRefined ----
    w_out = {}
    img = texture(320, 240, seed=3)
    for k, H in enumerate(warp_cases()):
        rc, (buf, dw, dh, dws) = R.image_projection_transform(img, H)
        w_out[f"ipt{k}"] = buf
        w_out[f"ipt{k}_dims"] = np.array([dw, dh, dws], np.int32)
    gray = np.ascontiguousarray(img[..., 1])
    rc, (buf, dw, dh, dws) = R.image_projection_transform(gray, warp_cases()[3])
    w_out["ipt_gray"] = buf
    w_out["ipt_gray_dims"] = np.array([dw, dh, dws], np.int32)
    imgs, h9s = mosaic_case()
    rc, (canvas, cw, ch, cws) = R.mosaic_images_refined(imgs, h9s)
    w_out["mosaic"] = canvas
    w_out["mosaic_dims"] = np.array([cw, ch, cws], np.int32)
    h9s2 = h9s.copy(); h9s2[2, 8] = 0
    rc, (canvas, cw, ch, cws) = R.mosaic_images_refined(imgs, h9s2)
    w_out["mosaic_skip"] = canvas
    w_out["mosaic_skip_dims"] = np.array([cw, ch, cws], np.int32)
    np.savez_compressed(os.path.join(HERE, "warp_golden.npz"), **w_out)
    # ---- blend path pieces the reference's own code produces: ResampleByOverlap decisions, chips + distance-map masks ----
    from tests.test_overlap import overlap_layouts, blend_cases
    b_out = {}
    for k, (w, h, h9) in enumerate(overlap_layouts(7)):
        b_out[f"keep{k}"] = R.resample_by_overlap(w, h, h9, 0.7)
    for tag, imgs, h9s in blend_cases():
        r = R.chips_and_masks(imgs, h9s, keep=np.ones(len(imgs), np.uint8), find_masks=True)
        b_out[f"{tag}_dims"] = np.array([r["cw"], r["ch"], len(r["chips"])], np.int32)
        for i in range(len(r["chips"])):
            b_out[f"{tag}_chip{i}"] = r["chip_imgs"][i]
            b_out[f"{tag}_mask{i}"] = r["masks"][i]
            b_out[f"{tag}_quad{i}"] = r["chips"][i]["quad"].view(np.uint32)
            b_out[f"{tag}_geom{i}"] = np.array([r["chips"][i][f] for f in ("x0", "y0", "w", "h", "img")], np.int32)
    np.savez_compressed(os.path.join(HERE, "blend_golden.npz"), **b_out)
    # ransac_polish_diverges.npz: the INPUTS (359 random correspondences, seed, distance) were found by scratch/soak_ransac.py on
    # the GPU box and are kept as they are; the expected outputs are the reference's, for sample_times 200 and 201
    f = os.path.join(HERE, "ransac_polish_diverges.npz")
    if os.path.exists(f):
        g = dict(np.load(f))
        out = {k: g[k] for k in ("p1", "p2", "dist", "seed")}
        for st in (200, 201):
            ok, i1, i2, H = R.ransac2d(g["p1"], g["p2"], float(g["dist"]), st, int(g["seed"]))
            out[f"ok{st}"] = np.int32(ok); out[f"i1_{st}"] = i1; out[f"i2_{st}"] = i2; out[f"H{st}"] = H
        np.savez_compressed(f, **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
