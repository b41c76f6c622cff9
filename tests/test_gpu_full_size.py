"""GPU: BASELINE configs C4 and C5 at their OWN size on one MI355X, under the oracle (VERDICT r02 "next" #1).

  C4  500 frames of 4000x3000, the reference's pair window j in (i, i+182) (MosaicWithoutPos.cpp:5083-5084) = all 74 029 pairs on
      one GPU.  Checked against the oracle: the features of 16 random frames (every keypoint field, every descriptor byte), EVERY
      accepted pair record (n_selected, n_in, inlier lists, H bit patterns) and 2 000 random rejected ones -- oracle.match_pair
      runs on the GPU's features of all 500 frames, on the box's host threads.
  C5  2000 frames of 4000x3000 resident in HBM (72 GB), global transforms whose bounding box is a 20000 x 20000-class canvas
      (SURVEY 8d).  MosaicImagesRefined: 8 stripes (one per GPU of the node) == the whole canvas, byte for byte, and
      4 random 1024^2 windows == the oracle's image-after-image overwrite.  LaplacianPyramidBlending (MosaicImage.cpp:2205-2510)
      through mi355_mosaic_blended_dev with ALL chips, their masks and the blender's pyramids co-resident with the frames (180 GB):
      4 random 1024^2 windows == oracle chips -> FindMasksByDistMap -> multiband blend (oracle_blend.c).

The window checks render with the oracle only what can reach the window: the images (chips) that intersect it, in image order,
inside the full canvas geometry (the layout functions see all 2000 transforms).  For the blend the window is cut with a 256 px
margin on a 32 px grid: a pixel of the 5-band result depends on chip pixels at most ~190 px away (REDUCE / EXPAND reach 2 px per
level, summed over the levels down and up), and a crop whose origin is a multiple of 2^5 keeps every level's sampling phase.
  C5 pair stage: the same 2000 resident frames through detect+describe and the reference's window: 345 529 pairs in the
      32 768-pair batches of the library, ~118 000 accepted.  Checked against the oracle with the C4 test's sampling: the features of 16
      random frames, 2 000 random accepted records + 1 000 random rejected ones (n_selected, n_in, inlier lists, H bit patterns).
Frames come from mi355_synth_frame_dev; everything is compared bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _threads():
    return max(1, min(192, (os.cpu_count() or 2) - 2))


def test_c4_full_size_one_gpu():
    import torch
    import imagemosaicing_amd as im
    from tests import oracle_lib as ol
    from tests.synth_survey import render_frames, host_image
    orc = ol.load_oracle_fast()
    ctx = im.Context(0)
    w, h, F = 4000, 3000, 500
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h)                  # 18 GB resident
    for k in range(F):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    pairs = im.pair_schedule(F, 182)
    assert len(pairs) == 74029                                                 # C4
    results = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")
    seed = 17
    torch.cuda.synchronize()                                                   # the zero fill runs on torch's stream, the library on the ctx's own
    ctx.MatchPairsDev(pairs, results.data_ptr(), 2.5, seed)
    ctx.synchronize()
    res = results.cpu().numpy().reshape(-1).view(im.PAIR_RESULT)
    assert np.array_equal(np.stack([res["i"], res["j"]], 1), pairs)
    feats = [ctx.GetFeatures(k) for k in range(F)]
    assert all(2000 <= len(f[0]) <= 2048 for f in feats)       # nfeatures + ties with the last one (retainBest)
    # (1) features of 16 random frames against oracle.sift
    rng = np.random.default_rng(4)
    pick = sorted(rng.choice(F, 16, replace=False).tolist())
    imgs = [host_image(frames, k, w, h, ws) for k in pick]
    ofe = ol.parallel_map(lambda a: orc.sift(a), imgs, threads=min(16, _threads()))
    for k, (okp, od) in zip(pick, ofe):
        kp, d = feats[k]
        assert len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)), f"C4 frame {k}: keypoints differ"
        assert np.array_equal(d.astype(np.uint8), od), f"C4 frame {k}: descriptors differ"
    # (2) every accepted record + 2000 random rejected ones against oracle.match_pair on the GPU's features
    acc = np.flatnonzero(res["accepted"] == 1)
    rej = np.flatnonzero(res["accepted"] == 0)
    assert 1500 < len(acc) < 4000, len(acc)                                    # adjacent + next-but-one + neighbouring rows of the serpentine
    check = np.concatenate([acc, rng.choice(rej, 2000, replace=False)])
    xy = [np.stack([f[0]["x"], f[0]["y"]], 1) for f in feats]
    d8 = [f[1].astype(np.uint8) for f in feats]

    def one(p):
        i, j = int(pairs[p][0]), int(pairs[p][1])
        return orc.match_pair(xy[i], d8[i], xy[j], d8[j], w, h, 2.5, seed)

    out = ol.parallel_map(one, check.tolist(), threads=_threads())
    bad = []
    for p, (nin, i1, i2, Ho, ns) in zip(check.tolist(), out):
        r = res[p]
        ok = ns == int(r["n_selected"]) and int(r["accepted"]) == int(nin > 30)
        if ok and nin > 30:
            ok = nin == int(r["n_in"]) and np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["b"][:nin], i2[:nin]) and \
                np.array_equal(r["H"].view(np.uint32), Ho.view(np.uint32))
        if not ok:
            bad.append((int(pairs[p][0]), int(pairs[p][1])))
    assert not bad, f"C4: {len(bad)} of {len(check)} checked records differ from the oracle, first {bad[:5]}"
    # the driver step on the full record set: one connected survey
    mp = im.results_to_match_pairs(res[acc])
    label = im.select_connected(mp, F)
    assert int(label.sum()) == F
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------------
def _bbox(h9, w, h):
    c = np.array([[0, 0, 1], [w - 1, 0, 1], [w - 1, h - 1, 1], [0, h - 1, 1]], np.float64).T
    p = h9.reshape(3, 3).astype(np.float64) @ c
    p = p[:2] / p[2]
    return p[0].min(), p[1].min(), p[0].max(), p[1].max()


def test_c5_full_size_canvas_and_blend():
    import torch
    import imagemosaicing_amd as im
    from tests import oracle_lib as ol
    from tests.synth_survey import affine3, host_image, block_layout
    orc = ol.load_oracle_fast()
    ctx = im.Context(0)
    w, h, F = 4000, 3000, 2000
    ws = (3 * w + 3) & ~3
    A = block_layout(F, w, h)
    rng = np.random.default_rng(8)
    frames = torch.empty((F, h * ws), dtype=torch.uint8, device="cuda")        # 72 GB resident
    for k in range(F):
        ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC5C5C5, k, 1 + rng.uniform(-0.05, 0.05), 2.0)
    ctx.synchronize()
    # global transforms (frame -> canvas): ground truth relative to frame 0, a small projective term on every third image, a few invalid
    h9 = np.zeros((F, 9), np.float32)
    A0i = np.linalg.inv(affine3(A[0]))
    for k in range(F):
        Hk = A0i @ affine3(A[k])
        if k % 3 == 1:
            Hk[2, 0], Hk[2, 1] = rng.normal(0, 2e-7), rng.normal(0, 2e-7)
        h9[k] = Hk.reshape(9)
    h9[[333, 1500], 8] = 0                                                     # skipped by the warp (MosaicWithoutPos.cpp:4646-4652)
    fptr = [frames[k].data_ptr() for k in range(F)]
    wv, hv, wsv = [w] * F, [h] * F, [ws] * F

    # ---- MosaicImagesRefined: 8 stripes == whole, windows == oracle --------------------------------------------------------
    cw, ch, cws, dG = im.mosaic_layout(wv, hv, h9)
    assert 19000 < cw < 22000 and 19000 < ch < 22000, (cw, ch)
    whole = torch.full((ch * cws,), 7, dtype=torch.uint8, device="cuda")
    stripes = torch.full((ch * cws,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                                                   # the fills run on torch's stream, the library on the ctx's own
    ctx.MosaicImagesRefinedDev(fptr, wv, hv, wsv, h9, whole.data_ptr(), cw, ch, cws)
    G = 8
    for r in range(G):
        row0 = (ch * r) // G
        ctx.MosaicImagesRefinedDev(fptr, wv, hv, wsv, h9, stripes.data_ptr(), cw, ch, cws, row0, (ch * (r + 1)) // G - row0)
    ctx.synchronize()
    assert torch.equal(whole, stripes), "C5: 8 canvas stripes differ from the whole canvas"
    del stripes
    boxes = [None if h9[k, 8] == 0 else _bbox(h9[k], w, h) for k in range(F)]
    minx = min(b[0] for b in boxes if b); miny = min(b[1] for b in boxes if b)
    extreme = {min((k for k in range(F) if boxes[k]), key=lambda k: boxes[k][0]), min((k for k in range(F) if boxes[k]), key=lambda k: boxes[k][1]),
               max((k for k in range(F) if boxes[k]), key=lambda k: boxes[k][2]), max((k for k in range(F) if boxes[k]), key=lambda k: boxes[k][3])}
    wins = [(int(rng.integers(0, cw - 1024)), int(rng.integers(0, ch - 1024))) for _ in range(4)]
    whole2d = whole.view(ch, cws)

    def refined_window(win):
        x0, y0 = win
        # images whose canvas bounding box (grown by 2 px) touches the window, plus the images that define the canvas box: the
        # oracle lays out the same canvas and writes the same bytes inside the window, in the same image order
        sub = sorted(set(k for k in range(F) if boxes[k] and boxes[k][0] - minx - 2 < x0 + 1024 and boxes[k][2] - minx + 2 > x0 and
                         boxes[k][1] - miny - 2 < y0 + 1024 and boxes[k][3] - miny + 2 > y0) | extreme)
        imgs = [host_image(frames, k, w, h, ws) for k in sub]
        rc, (ref, rw, rh, rws) = orc.mosaic_images_refined(imgs, h9[sub])
        assert rc == 0 and (rw, rh, rws) == (cw, ch, cws)
        return ref[y0:y0 + 1024, 3 * x0:3 * (x0 + 1024)].copy(), len(sub)

    for win, (ref, nsub) in zip(wins, ol.parallel_map(refined_window, wins, threads=4)):
        x0, y0 = win
        got = whole2d[y0:y0 + 1024, 3 * x0:3 * (x0 + 1024)].cpu().numpy()
        assert nsub > 20 and np.array_equal(got, ref), f"C5 canvas window {win}: {int((got != ref).sum())} bytes differ ({nsub} images)"
    del whole, whole2d

    # ---- LaplacianPyramidBlending, everything co-resident --------------------------------------------------------------------
    # vecAbandonInd: the reference's ResampleByOverlap(0.7) (MosaicImage.cpp:2069-2201; host function, == the reference's own code in
    # tests/test_overlap.py) keeps ALL 2000 images of this survey -- it only measures overlaps whose polygon has 3 or 4 corners, and
    # rectangles that differ by a small yaw intersect in 6 to 8.  So all ~1998 valid chips (3 B) and masks (1 B per chip pixel: 101 GB)
    # sit beside the 72 GB of frames and the blender's pyramids; the distance maps are not stored any more (warp.hip owner_kernel).
    keep = im.resample_by_overlap(wv, hv, h9, 0.7)
    assert keep[0] == 1 and keep[F - 1] == 1 and int(keep.sum()) == F
    band = 5
    out, bw, bh, bws = ctx.MosaicBlendedDev(fptr, wv, hv, wsv, h9, keep=keep, band=band)     # torch uint8 [bh, bws] in HBM
    ctx.synchronize()
    # oracle geometry of ALL chips (transforms only), pixels of the chips that reach the window only
    w_a, h_a = np.array(wv, np.int32), np.array(hv, np.int32)
    chips, lw, lh, ldG = orc.chip_layout(w_a, h_a, h9, keep)
    assert (lw, lh) == (bw, bh)
    M, AL = 256, 1 << band

    def blended_window(win):
        x0, y0 = win
        cx0, cy0 = max(0, (x0 - M) // AL * AL), max(0, (y0 - M) // AL * AL)
        cx1, cy1 = min(bw, x0 + 1024 + M), min(bh, y0 + 1024 + M)
        sub = [c for c in chips if c["x0"] < cx1 and c["x0"] + c["w"] > cx0 and c["y0"] < cy1 and c["y0"] + c["h"] > cy0]
        cm = ol.parallel_map(lambda c: orc.chip_warp(host_image(frames, int(c["img"]), w, h, ws), h9[int(c["img"])], ldG, c), sub, threads=min(48, _threads()))
        cimgs, masks = [c_ for c_, _ in cm], [m_ for _, m_ in cm]
        # ownership inside the crop only: chip origins relative to the crop, rect = the crop (pixels outside stay 0 = not owned)
        shifted = np.array(sub, ol.CHIPINFO)
        shifted["x0"] -= cx0; shifted["y0"] -= cy0
        orc.find_masks_by_distmap(masks, shifted, cx1 - cx0, cy1 - cy0)
        # cut every chip to the crop
        cut_info, cut_chips, cut_masks = [], [], []
        for c, chip, mask in zip(shifted, cimgs, masks):
            ax0, ay0 = max(0, -int(c["x0"])), max(0, -int(c["y0"]))
            ax1, ay1 = min(int(c["w"]), cx1 - cx0 - int(c["x0"])), min(int(c["h"]), cy1 - cy0 - int(c["y0"]))
            cwid, chei = ax1 - ax0, ay1 - ay0
            cc = np.zeros((chei, (cwid * 3 + 3) & ~3), np.uint8); cc[:, :cwid * 3] = chip[ay0:ay1, 3 * ax0:3 * ax1]
            mm = np.zeros((chei, (cwid + 3) & ~3), np.uint8); mm[:, :cwid] = mask[ay0:ay1, ax0:ax1]
            cut_info.append((int(c["x0"]) + ax0, int(c["y0"]) + ay0, cwid, chei)); cut_chips.append(cc); cut_masks.append(mm)
        info = np.zeros(len(cut_info), ol.CHIPINFO)
        for q, (a, b, c_, d) in enumerate(cut_info):
            info[q]["x0"], info[q]["y0"], info[q]["w"], info[q]["h"], info[q]["img"] = a, b, c_, d, q
        ref, nb = orc.multiband_blend(info, cut_chips, cut_masks, cx1 - cx0, cy1 - cy0, band=band)
        assert nb == band
        return ref[y0 - cy0:y0 - cy0 + 1024, 3 * (x0 - cx0):3 * (x0 - cx0 + 1024)].copy(), len(sub)

    bwins = [(int(rng.integers(0, bw - 1024)), int(rng.integers(0, bh - 1024))) for _ in range(4)]
    for win, (ref, nsub) in zip(bwins, [blended_window(bw_) for bw_ in bwins]):
        x0, y0 = win
        got = out[y0:y0 + 1024, 3 * x0:3 * (x0 + 1024)].cpu().numpy()
        assert nsub >= 2 and np.array_equal(got, ref), f"C5 blend window {win}: {int((got != ref).sum())} bytes differ ({nsub} chips)"
    # ---- the same canvas as the 8 stripes 8 ranks would blend (mi355_mosaic_blended_rows_dev): byte for byte the whole one ----
    import torch
    edges = np.linspace(0, bh, 9).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        part, _, _, _ = ctx.MosaicBlendedDev(fptr, wv, hv, wsv, h9, keep=keep, band=band, row0=int(a), rows=int(b - a))
        assert torch.equal(part, out[int(a):int(b)]), f"C5 blend stripe rows {a}..{b - 1} differs from the whole canvas"
        del part
    ctx.close()


def test_c5_pair_stage_full_size():
    """MosaicWithoutPos.cpp:5083-5084 on 2000 frames: j in (i, i + 182) -> 345 529 pairs, on the GPU's own features of all frames"""
    import torch
    import imagemosaicing_amd as im
    from tests import oracle_lib as ol
    from tests.synth_survey import host_image, block_layout
    orc = ol.load_oracle_fast()
    ctx = im.Context(0)
    w, h, F = 4000, 3000, 2000
    ws = (3 * w + 3) & ~3
    A = block_layout(F, w, h)
    rng = np.random.default_rng(8)
    frames = torch.empty((F, h * ws), dtype=torch.uint8, device="cuda")        # 72 GB resident
    for k in range(F):
        ctx.SynthFrameDev(frames[k].data_ptr(), w, h, ws, A[k], 0xC5C5C5, k, 1 + rng.uniform(-0.05, 0.05), 2.0)
    ctx.synchronize()
    for k in range(F):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    pairs = im.pair_schedule(F, 182)
    assert len(pairs) == 345529                                                # C5
    results = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")   # 3.3 GB
    seed = 23
    torch.cuda.synchronize()
    ctx.MatchPairsDev(pairs, results.data_ptr(), 2.5, seed)
    ctx.synchronize()
    # the record fields that decide the sample, without pulling 3.3 GB through PCIe: i, j, accepted of every record, whole records of the sample
    head = results[:, :16].cpu().numpy().copy()
    offs = {n: im.PAIR_RESULT.fields[n][1] for n in ("i", "j", "accepted")}
    col = lambda n: results[:, offs[n]:offs[n] + 4].contiguous().view(torch.int32).reshape(-1).cpu().numpy()
    assert np.array_equal(np.stack([col("i"), col("j")], 1), pairs)
    accepted = col("accepted")
    acc, rej = np.flatnonzero(accepted == 1), np.flatnonzero(accepted == 0)
    assert 80000 < len(acc) < 160000, len(acc)
    check = np.concatenate([rng.choice(acc, 2000, replace=False), rng.choice(rej, 1000, replace=False)])
    res = results[torch.from_numpy(check).cuda()].cpu().numpy().reshape(-1).view(im.PAIR_RESULT)
    del head
    used = sorted(set(pairs[check].reshape(-1).tolist()))
    feats = {k: ctx.GetFeatures(k) for k in used}
    assert all(2000 <= len(f[0]) <= 2048 for f in feats.values())
    # (1) features of 16 random frames against oracle.sift
    pick = sorted(rng.choice(F, 16, replace=False).tolist())
    imgs = [host_image(frames, k, w, h, ws) for k in pick]
    ofe = ol.parallel_map(lambda a: orc.sift(a), imgs, threads=min(16, _threads()))
    for k, (okp, od) in zip(pick, ofe):
        kp, d = ctx.GetFeatures(k)
        assert len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)), f"C5 frame {k}: keypoints differ"
        assert np.array_equal(d.astype(np.uint8), od), f"C5 frame {k}: descriptors differ"
    # (2) the sampled records against oracle.match_pair on the GPU's features
    xy = {k: np.stack([f[0]["x"], f[0]["y"]], 1) for k, f in feats.items()}
    d8 = {k: f[1].astype(np.uint8) for k, f in feats.items()}

    def one(q):
        i, j = int(pairs[check[q]][0]), int(pairs[check[q]][1])
        return orc.match_pair(xy[i], d8[i], xy[j], d8[j], w, h, 2.5, seed)

    out = ol.parallel_map(one, list(range(len(check))), threads=_threads())
    bad = []
    for q, (nin, i1, i2, Ho, ns) in enumerate(out):
        r = res[q]
        ok = ns == int(r["n_selected"]) and int(r["accepted"]) == int(nin > 30)
        if ok and nin > 30:
            ok = nin == int(r["n_in"]) and np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["b"][:nin], i2[:nin]) and \
                np.array_equal(r["H"].view(np.uint32), Ho.view(np.uint32))
        if not ok:
            bad.append((int(r["i"]), int(r["j"])))
    assert not bad, f"C5: {len(bad)} of {len(check)} checked records differ from the oracle, first {bad[:5]}"
    ctx.close()
