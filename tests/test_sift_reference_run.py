"""SIFT against the reference's ONE committed run (the only output of OpenCV 2.4.0's SIFT that exists anywhere in the reference tree):
Release/feature_temp/matchPairs.match holds the inlier keypoints cv::SIFT produced on Release/test_data/DSC00004..23.JPG.  Two of
those frames are committed as fixtures (data files of the reference: tests/golden/DSC00004.JPG, DSC00005.JPG = images 0 and 1 of
the run).  SIFT parity stays UNPINNED at the bit level (no OpenCV source), but this measurement decides the one free choice that
shows up as a systematic effect -- how the base image is doubled (oracle_sift.c orc_sift) -- and keeps the agreement from regressing:

  * CPU: >= 60 % of the reference's keypoints of the two frames have an oracle keypoint within 0.1 px (71 % measured over ten
    frames of the run; pixel-centre aligned doubling gives 0 % within 0.1 px before and 52 % after removing its (0.25, 0.25) px
    bias), with no systematic offset left;
  * GPU: the HIP path equals the oracle bit for bit on these real photographs, features and the pair record."""
import os

import numpy as np
import pytest

from tests.golden_util import GOLD, load_match_pairs

PIL = pytest.importorskip("PIL.Image")


def frames():
    return [np.ascontiguousarray(np.array(PIL.open(os.path.join(GOLD, "DSC%05d.JPG" % (4 + k))).convert("RGB"))[:, :, ::-1]) for k in range(2)]


def reference_keypoints(k):
    mp = load_match_pairs()
    pts = {}
    for side in ("a", "b"):
        m = mp[side + "i"] == k
        for i, x, y in zip(mp[side + "id"][m], mp[side + "x"][m], mp[side + "y"][m]):
            pts[int(i)] = (float(x), float(y))
    return np.array(list(pts.values()), np.float64)


def test_sift_keypoints_agree_with_the_reference_run():
    from tests import oracle_lib as ol
    orc = ol.load_oracle_fast()
    dists, resid = [], []
    for k, img in enumerate(frames()):
        assert img.shape == (750, 1000, 3)
        kp, _ = orc.sift(img, nfeatures=30000, max_kp=30000)          # the run kept more than 2000 keypoints (ids up to 3130): compare with all
        P = np.stack([kp["x"], kp["y"]], 1).astype(np.float64)
        R = reference_keypoints(k)
        assert len(R) > 400
        D2 = ((R[:, None, :] - P[None, :, :]) ** 2).sum(-1)
        j = D2.argmin(1)
        d = np.sqrt(D2.min(1))
        dists.append(d)
        resid.append((R - P[j])[d < 0.5])
    d = np.concatenate(dists); r = np.concatenate(resid)
    assert (d < 0.1).mean() >= 0.60, (d < 0.1).mean()
    assert (d < 0.5).mean() >= 0.80, (d < 0.5).mean()
    assert np.abs(r.mean(0)).max() < 0.03, r.mean(0)                   # no systematic offset (pixel-centre doubling: 0.25 px in x and y)


@pytest.mark.gpu
def test_sift_gpu_equals_oracle_on_the_reference_frames():
    import imagemosaicing_amd as im
    from tests import oracle_lib as ol
    orc = ol.load_oracle_fast()
    ctx = im.Context(0)
    fs = frames()
    feats = []
    for k, img in enumerate(fs):
        kp, desc = ctx.SiftExtract(k, img)
        okp, odesc = orc.sift(img)
        assert len(kp) == len(okp) == 2000
        assert np.array_equal(kp.view(np.uint8), okp.view(np.uint8)), f"frame {k}: keypoints differ"
        assert np.array_equal(desc.astype(np.uint8), odesc), f"frame {k}: descriptors differ"
        feats.append((okp, odesc))
    r = ctx.MatchPairs([(0, 1)], 2.5, 1)[0]
    (k1, d1), (k2, d2) = feats
    nin, i1, i2, Ho, ns = orc.match_pair(np.stack([k1["x"], k1["y"]], 1), d1, np.stack([k2["x"], k2["y"]], 1), d2, 1000, 750, 2.5, 1)
    assert int(r["accepted"]) == 1 and int(r["n_in"]) == nin > 100 and int(r["n_selected"]) == ns
    assert np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["H"].view(np.uint32), Ho.view(np.uint32))
    # the pairwise motion agrees with the reference's global solution for image 1 (tran0.txt row 1: tx 11.58, ty -100.62) to a few pixels
    assert abs(float(r["H"][2]) - 11.58) < 6 and abs(float(r["H"][5]) + 100.62) < 6
    ctx.close()
