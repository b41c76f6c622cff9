"""How far is the SIFT orientation / descriptor arithmetic the product implements (oracle mode 0: order-free histogram sums, polynomial exp /
sincos, fused atan2 polynomial) from the closest restatement of the reference's binary that can be made here (oracle mode 1,
oracle_sift.c orc_sift_set_mode: cv::exp's SSE blocks of eight + double tail over the COMPACTED in-window samples, fastAtan2 with separately
rounded products and sums, float histogram sums in scan order, the C library's cosf / sinf / powf)?  Nothing in the reference tree holds an
angle or a descriptor of its run, so neither mode can be compared with the binary's output directly; what can be measured is
  (i)  how many angles and descriptor bytes move between the two modes on the reference's own photographs, and
  (ii) whether the reference's committed inlier correspondences (found under the BINARY's descriptors) are nearest neighbours more often
       under one mode's descriptors than under the other's.
Measured on the 20 frames of the reference's run (build container; the two committed frames elsewhere) -- the numbers are asserted below and
quoted in DESIGN.md section 2a."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tests.golden_util import load_match_pairs
from tests.test_sift_reference_run import frame, available_frames, REF_DATA

DLL = "/root/reference/code/MosaicingCode/Release/opencv_core240.dll"


def _sift(orc, k, mode):
    orc.L.orc_sift_set_mode(mode)
    try:
        return orc.sift(frame(k), nfeatures=0, max_kp=30000)
    finally:
        orc.L.orc_sift_set_mode(0)


def test_cv_exp_restated_table_and_block_tail_structure():
    from tests import oracle_lib as ol
    orc = ol.load_oracle()
    orc.L.orc_cv_exp_table.restype = C.POINTER(C.c_double)
    tab = np.array([orc.L.orc_cv_exp_table()[k] for k in range(64)])
    if os.path.exists(DLL):          # the table the binary holds (file offset 0x1693c0 = 1016a7c0), byte for byte
        raw = open(DLL, "rb").read()
        at = raw.find(struct.pack("<d", 0.9670371139572337719125840413672004409288e-2))
        assert at > 0 and np.array_equal(np.frombuffer(raw[at:at + 512], np.float64), tab)
    # values: within 2 ulp of exp() everywhere; an element's value depends on whether it sits in a block of eight or in the tail
    x = -np.abs(np.random.default_rng(1).normal(0, 3, 1003)).astype(np.float32)
    y = np.zeros_like(x)
    orc.L.orc_cv_exp32f(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x))
    want = np.exp(x.astype(np.float64))
    assert np.abs(y / want - 1).max() < 3e-7
    y2 = np.zeros(8, np.float32); xs = np.ascontiguousarray(x[-8:])          # the last 3 of 1003 went through the tail; as a block of eight they take the SSE path
    orc.L.orc_cv_exp32f(xs.ctypes.data_as(C.c_void_p), y2.ctypes.data_as(C.c_void_p), 8)
    assert np.abs(y2 / want[-8:] - 1).max() < 3e-7


def test_orientation_and_descriptor_bytes_between_the_products_definition_and_the_binarys_order():
    from tests import oracle_lib as ol
    orc = ol.load_oracle_fast()
    ks = available_frames()
    mp = load_match_pairs()
    f0 = {k: _sift(orc, k, 0) for k in ks}
    f1 = {k: _sift(orc, k, 1) for k in ks}
    n_kp = n_same_kp = 0
    ang_diff = []; byte_diff = []; byte_moved = 0; byte_tot = 0; desc_moved = 0
    for k in ks:
        (ka, da), (kb, db) = f0[k], f1[k]
        n_kp += len(ka)
        if len(ka) != len(kb) or not (np.array_equal(ka["x"], kb["x"]) and np.array_equal(ka["y"], kb["y"])):
            # a different number of orientation peaks somewhere shifts the list: align on (x, y, octave word) and the rank among equals
            key = lambda kp: [(float(a), float(b), int(c)) for a, b, c in zip(kp["x"], kp["y"], kp["octave"])]
            from collections import defaultdict
            ia, ib = defaultdict(list), defaultdict(list)
            for i, t in enumerate(key(ka)): ia[t].append(i)
            for i, t in enumerate(key(kb)): ib[t].append(i)
            pa = [i for t in ia if len(ia[t]) == len(ib.get(t, [])) for i in ia[t]]
            pb = [i for t in ia if len(ia[t]) == len(ib.get(t, [])) for i in ib[t]]
        else:
            pa = pb = list(range(len(ka)))
        pa, pb = np.array(pa, int), np.array(pb, int)
        n_same_kp += len(pa)
        ang_diff.append(np.abs(ka["angle"][pa].astype(np.float64) - kb["angle"][pb].astype(np.float64)))
        d = np.abs(da[pa].astype(np.int16) - db[pb].astype(np.int16))
        byte_diff.append(d.max(1)); byte_moved += int((d != 0).sum()); byte_tot += d.size; desc_moved += int((d != 0).any(1).sum())
    ang = np.concatenate(ang_diff); bd = np.concatenate(byte_diff)
    ang = np.minimum(ang, 360.0 - ang)
    # nearest-neighbour agreement with the reference's inlier correspondences under either mode's descriptors
    def agreement(feats):
        hit = tot = 0
        misses = []
        for (a, b) in sorted(set(zip(mp["ai"].tolist(), mp["bi"].tolist()))):
            if a not in feats or b not in feats:
                continue
            m = (mp["ai"] == a) & (mp["bi"] == b)
            da_, db_ = feats[a][1].astype(np.int32), feats[b][1].astype(np.int32)
            q = da_[mp["aid"][m]]
            d2 = (q * q).sum(1)[:, None] + (db_ * db_).sum(1)[None, :] - 2 * q @ db_.T
            ok = d2.argmin(1) == mp["bid"][m]
            hit += int(ok.sum()); tot += int(m.sum())
            misses += [(a, b, int(i)) for i in np.flatnonzero(~ok)]
        return hit, tot, misses
    h0, t0, m0 = agreement(f0)
    h1, t1, m1 = agreement(f1)
    report = {"frames": len(ks), "keypoints": n_kp, "keypoints_compared": n_same_kp,
              "angles_that_differ": int((ang != 0).sum()), "angle_diff_max_deg": float(ang.max()), "angle_diff_gt_0.01deg": int((ang > 0.01).sum()),
              "descriptors_with_a_moved_byte": desc_moved, "bytes_moved": byte_moved, "bytes": byte_tot, "max_byte_step": int(bd.max()),
              "descriptors_moved_by_more_than_1": int((bd > 1).sum()),
              "nn_agreement_mode0": [h0, t0], "nn_agreement_mode1": [h1, t1], "misses_mode0": len(m0), "misses_mode1": len(m1),
              "misses_common": len(set(m0) & set(m1))}
    print("\nSIFT definition (mode 0) vs the binary's order (mode 1):", report)
    out = os.environ.get("MI355_SIFT_MODES_REPORT")
    if out:
        import json
        json.dump(report, open(out, "w"), indent=1)
    # the detector is common to both modes: same keypoint locations except where the number of orientation peaks changes
    assert n_same_kp >= 0.995 * n_kp
    # the orientation moves by a rounding's worth for nearly every keypoint (different exp / atan2 roundings) but not by more than the
    # histogram's resolution allows; a few keypoints change their number of peaks (compared keypoints < all keypoints)
    assert float(np.median(ang)) < 0.01 and float(np.percentile(ang, 99.9)) < 1.0
    # descriptor bytes: most moved bytes move by one level
    assert byte_moved < 0.12 * byte_tot and int((bd <= 1).sum()) > 0.97 * len(bd)
    # neither mode is closer to the binary's descriptors by the only evidence there is
    assert t0 == t1 and t0 >= (5918 if len(ks) == 20 else 250)
    assert h0 / t0 >= 0.99 and h1 / t1 >= 0.99 and abs(h0 - h1) <= max(3, t0 // 1000)
