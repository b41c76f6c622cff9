"""ctypes bindings of the CHECKERS: oracle/liboracle.so (CPU restatement) and
oracle/_ref/libref_oracle.so (the reference's own code, built by oracle/ref/build_ref.sh).

Test infrastructure only -- never imported by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")

SFPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("id", "<i4")])
KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CHIPINFO = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("w", "<i4"), ("h", "<i4"), ("img", "<i4"),
                     ("sx", "<f4"), ("sy", "<f4"), ("quad", "<f4", (8,))])

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
u8p = C.POINTER(C.c_uint8)


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def sfpoints(xy, ids=None):
    xy = np.asarray(xy, np.float32)
    out = np.zeros(len(xy), SFPOINT)
    out["x"] = xy[:, 0]
    out["y"] = xy[:, 1]
    out["id"] = np.arange(len(xy)) if ids is None else ids
    return out


def build_oracle():
    so = os.path.join(ORC_DIR, "liboracle.so")
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".c", ".h"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORC_DIR, "liboracle.so", "liboracle_v3.so"])
    return so


class Oracle:
    def __init__(self, path):
        self.L = C.CDLL(path)
        L = self.L
        L.orc_rand.restype = C.c_int

    # ---- homography -------------------------------------------------------
    def inverse_matrix(self, a, eps):
        a = np.ascontiguousarray(a, np.float32)
        n = a.shape[0]
        out = np.zeros((n, n), np.float32)
        rc = self.L.orc_inverse_matrix(_p(a), n, _p(out), C.c_float(eps))
        return rc, out

    def solve_homography(self, p1, p2):
        H = np.zeros(9, np.float32)
        rc = self.L.orc_solve_homography(_p(p1), _p(p2), len(p1), _p(H))
        return rc, H

    def nlls(self, p1, p2, H0, stop=1e-10):
        H = np.zeros(9, np.float32)
        H0 = np.ascontiguousarray(H0, np.float32)
        rc = self.L.orc_nlls_projection2(_p(p1), _p(p2), len(p1), _p(H), _p(H0), C.c_float(stop))
        return rc, H

    def rand_stream(self, seed, n):
        st = (C.c_int32 * 40)()
        self.L.orc_srand(st, C.c_uint(seed))
        return np.array([self.L.orc_rand(st) for _ in range(n)], np.int64)

    def ransac2d(self, p1, p2, dist, sample_times, seed):
        n = len(p1)
        i1 = np.zeros(max(n, 1), SFPOINT)
        i2 = np.zeros(max(n, 1), SFPOINT)
        nin = C.c_int(0)
        H = np.zeros(9, np.float32)
        ok = self.L.orc_ransac2d(_p(p1), _p(p2), n, C.c_float(dist), sample_times, C.c_uint(seed),
                                 _p(i1), _p(i2), C.byref(nin), _p(H))
        return ok, i1[:nin.value].copy(), i2[:nin.value].copy(), H

    # ---- selection / matching --------------------------------------------
    def select(self, matches, kp1, kp2, nMatch, w, h, gx=3, gy=3):
        matches = np.ascontiguousarray(matches, np.int32)
        kp1 = np.ascontiguousarray(kp1, np.float32)
        kp2 = np.ascontiguousarray(kp2, np.float32)
        n = len(matches)
        o1 = np.zeros(max(n, 1), SFPOINT)
        o2 = np.zeros(max(n, 1), SFPOINT)
        no = C.c_int(0)
        self.L.orc_select_match_pairs(_p(matches), n, _p(kp1), _p(kp2), nMatch, w, h, gx, gy, _p(o1), _p(o2), C.byref(no))
        return o1[:no.value].copy(), o2[:no.value].copy()

    def bf_match(self, d1, d2):
        d1 = np.ascontiguousarray(d1, np.uint8)
        d2 = np.ascontiguousarray(d2, np.uint8)
        n1 = len(d1)
        idx = np.zeros(n1, np.int32)
        b1 = np.zeros(n1, np.int32)
        b2 = np.zeros(n1, np.int32)
        self.L.orc_bf_match(_p(d1), n1, _p(d2), len(d2), _p(idx), _p(b1), _p(b2))
        return idx, b1, b2

    def sort_matches(self, idx, d2):
        out = np.zeros((len(idx), 2), np.int32)
        self.L.orc_sort_matches(_p(np.ascontiguousarray(idx, np.int32)), _p(np.ascontiguousarray(d2, np.int32)), len(idx), _p(out))
        return out

    def match_pair_ratio(self, kp1, d1, kp2, d2, w, h, dist, seed, ratio):
        kp1 = np.ascontiguousarray(kp1, np.float32)
        kp2 = np.ascontiguousarray(kp2, np.float32)
        d1 = np.ascontiguousarray(d1, np.uint8)
        d2 = np.ascontiguousarray(d2, np.uint8)
        n1 = len(kp1)
        i1 = np.zeros(max(n1, 1), SFPOINT)
        i2 = np.zeros(max(n1, 1), SFPOINT)
        H = np.zeros(9, np.float32)
        ns = C.c_int(0)
        self.L.orc_match_pair_ratio.restype = C.c_int
        nin = self.L.orc_match_pair_ratio(_p(kp1), _p(d1), n1, _p(kp2), _p(d2), len(kp2), w, h, C.c_float(dist), C.c_uint(seed), C.c_float(ratio),
                                          _p(i1), _p(i2), _p(H), C.byref(ns))
        return nin, i1, i2, H, ns.value

    def match_pair(self, kp1, d1, kp2, d2, w, h, dist, seed):
        kp1 = np.ascontiguousarray(kp1, np.float32)
        kp2 = np.ascontiguousarray(kp2, np.float32)
        d1 = np.ascontiguousarray(d1, np.uint8)
        d2 = np.ascontiguousarray(d2, np.uint8)
        n1 = len(kp1)
        i1 = np.zeros(max(n1, 1), SFPOINT)
        i2 = np.zeros(max(n1, 1), SFPOINT)
        H = np.zeros(9, np.float32)
        ns = C.c_int(0)
        self.L.orc_match_pair.restype = C.c_int
        nin = self.L.orc_match_pair(_p(kp1), _p(d1), n1, _p(kp2), _p(d2), len(kp2), w, h, C.c_float(dist), C.c_uint(seed),
                                    _p(i1), _p(i2), _p(H), C.byref(ns))
        return nin, i1, i2, H, ns.value

    # ---- warps -------------------------------------------------------------
    def image_projection_transform(self, img, h9):
        img = np.ascontiguousarray(img)
        hh, ws = img.shape[0], img.strides[0]
        ch = img.shape[2] if img.ndim == 3 else 1
        w = img.shape[1]
        h9 = np.ascontiguousarray(h9, np.float32)
        dst = C.c_void_p()
        dw, dh, dws = C.c_int(), C.c_int(), C.c_int()
        rc = self.L.orc_image_projection_transform(_p(img), w, hh, ws, ch, _p(h9), C.byref(dst), C.byref(dw), C.byref(dh), C.byref(dws))
        if rc != 0:
            return rc, None
        buf = np.ctypeslib.as_array(C.cast(dst, u8p), shape=(dh.value, dws.value)).copy()
        self.L.orc_free(dst)
        return rc, (buf, dw.value, dh.value, dws.value)

    def mosaic_images_refined(self, imgs, h9s):
        n = len(imgs)
        imgs = [np.ascontiguousarray(i) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        cw, ch, cws = C.c_int(), C.c_int(), C.c_int()
        rc = self.L.orc_mosaic_images_refined(ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), None, C.byref(cw), C.byref(ch), C.byref(cws))
        if rc != 0:
            return rc, None
        canvas = np.zeros((ch.value, cws.value), np.uint8)
        rc = self.L.orc_mosaic_images_refined(ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(canvas), C.byref(cw), C.byref(ch), C.byref(cws))
        return rc, (canvas, cw.value, ch.value, cws.value)

    def multiband_blend(self, chips, chip_imgs, masks, cw, ch, band=5):
        n = len(chip_imgs)
        ci = [np.ascontiguousarray(c, np.uint8) for c in chip_imgs]
        mi = [np.ascontiguousarray(m, np.uint8) for m in masks]
        cp = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in ci])
        mp = (C.c_void_p * max(n, 1))(*[m.ctypes.data for m in mi])
        x0 = np.array([int(c["x0"]) for c in chips], np.int32); y0 = np.array([int(c["y0"]) for c in chips], np.int32)
        w = np.array([int(c["w"]) for c in chips], np.int32); h = np.array([int(c["h"]) for c in chips], np.int32)
        ows = (cw * 3 + 3) & ~3
        out = np.zeros((ch, ows), np.uint8)
        self.L.orc_multiband_blend.restype = C.c_int
        nb = self.L.orc_multiband_blend(cp, mp, _p(x0), _p(y0), _p(w), _p(h), n, int(cw), int(ch), int(band), _p(out))
        return out, nb

    def chip_layout(self, w, h, h9s, keep=None):
        """geometry of the chips alone (MosaicImage.cpp:2233-2343): (chips, cw, ch, dG)"""
        w = np.ascontiguousarray(w, np.int32); h = np.ascontiguousarray(h, np.int32)
        n = len(w)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep = np.ones(n, np.uint8) if keep is None else np.ascontiguousarray(keep, np.uint8)
        chips = np.zeros(n, CHIPINFO)
        cw, ch = C.c_int(), C.c_int()
        dG = np.zeros(2, np.float32)
        self.L.orc_chip_layout.restype = C.c_int
        nv = self.L.orc_chip_layout(_p(w), _p(h), n, _p(h9s), _p(keep), C.byref(cw), C.byref(ch), _p(dG), _p(chips))
        return chips[:nv].copy(), cw.value, ch.value, dG

    def chip_warp(self, img, h9, dG, c):
        """one chip + its validity mask (MosaicImage.cpp:2343-2448)"""
        img = np.ascontiguousarray(img)
        cws = (int(c["w"]) * 3 + 3) & ~3
        mws = (int(c["w"]) + 3) & ~3
        chip = np.zeros((int(c["h"]), cws), np.uint8)
        mask = np.zeros((int(c["h"]), mws), np.uint8)
        ci = np.array([c], CHIPINFO)
        h9 = np.ascontiguousarray(h9, np.float32)
        rc = self.L.orc_chip_warp(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(h9), _p(dG), _p(ci), _p(chip), cws, _p(mask), mws)
        assert rc == 0
        return chip, mask

    def find_masks_by_distmap(self, masks, chips, rect_w, rect_h):
        """FindMasksByDistMap (MosaicImage.cpp:1761-1881) in place on `masks`; ownership is decided for the pixels of the
        rectangle [0, rect_w) x [0, rect_h) in the chips' coordinate system"""
        nv = len(masks)
        chips = np.ascontiguousarray(chips, CHIPINFO)
        mp = (C.c_void_p * nv)(*[m.ctypes.data for m in masks])
        mws = np.array([m.strides[0] for m in masks], np.int32)
        self.L.orc_find_masks_by_distmap(mp, _p(mws), _p(chips), nv, int(rect_w), int(rect_h))

    def chips_and_masks(self, imgs, h9s, keep=None, find_masks=True):
        imgs = [np.ascontiguousarray(i) for i in imgs]
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        chips, cw, ch, dG = self.chip_layout(w, h, h9s, keep)
        cimgs, masks = [], []
        for c in chips:
            k = int(c["img"])
            chip, mask = self.chip_warp(imgs[k], h9s[k], dG, c)
            cimgs.append(chip)
            masks.append(mask)
        valid = [m.copy() for m in masks]
        if find_masks and len(chips):
            self.find_masks_by_distmap(masks, chips, cw, ch)
        return dict(cw=cw, ch=ch, dG=dG, chips=chips, chip_imgs=cimgs, valid=valid, masks=masks)

    def sift(self, bgr, nfeatures=2000, max_kp=None):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        max_kp = max_kp or (max(nfeatures, 2048) if nfeatures > 0 else nfeatures)     # nfeatures + ties with the last one (retainBest), up to the feature record's 2048
        kp = np.zeros(max_kp, KEYPOINT)
        desc = np.zeros((max_kp, 128), np.uint8)
        self.L.orc_sift.restype = C.c_int
        n = self.L.orc_sift(_p(bgr), w, h, bgr.strides[0], nfeatures, _p(kp), _p(desc), max_kp)
        return kp[:n].copy(), desc[:n].copy()


    # ---- SURF variant (row f4) ---------------------------------------------------------------------
    def surf(self, bgr, hessian=50.0, max_kp=4096):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        kp = np.zeros(max_kp, KEYPOINT)
        desc = np.zeros((max_kp, 128), np.float32)
        self.L.orc_surf.restype = C.c_int
        n = self.L.orc_surf(_p(bgr), w, h, bgr.strides[0], C.c_float(hessian), _p(kp), _p(desc), max_kp)
        return kp[:n].copy(), desc[:n].copy()

    def bf_match_f32(self, d1, d2):
        d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
        idx = np.zeros(len(d1), np.int32); dist = np.zeros(len(d1), np.float32)
        self.L.orc_bf_match_f32(_p(d1), len(d1), _p(d2), len(d2), _p(idx), _p(dist))
        return idx, dist

    def select_by_distance(self, idx, dist, kp1xy, kp2xy, match_dist=0.5, max_features=200):
        idx = np.ascontiguousarray(idx, np.int32); dist = np.ascontiguousarray(dist, np.float32)
        kp1xy = np.ascontiguousarray(kp1xy, np.float32); kp2xy = np.ascontiguousarray(kp2xy, np.float32)
        o1 = np.zeros(max(len(idx), 1), SFPOINT); o2 = np.zeros(max(len(idx), 1), SFPOINT)
        self.L.orc_select_by_distance.restype = C.c_int
        n = self.L.orc_select_by_distance(_p(idx), _p(dist), len(idx), _p(kp1xy), _p(kp2xy), C.c_float(match_dist), int(max_features), _p(o1), _p(o2))
        return o1[:n].copy(), o2[:n].copy()

    def surf_match_pair(self, f1, f2, ransac_dist=2.5, seed=1, match_dist=0.5, max_features=200, min_inliers=18):
        """the j-loop body of GetMatchedPairsOneToAllSurf (MosaicWithoutPos.cpp:5389-5517): returns (n_in or 0, in1, in2, H, n_selected)"""
        (k1, d1), (k2, d2) = f1, f2
        idx, dist = self.bf_match_f32(d1, d2)
        s1, s2 = self.select_by_distance(idx, dist, np.stack([k1["x"], k1["y"]], 1), np.stack([k2["x"], k2["y"]], 1), match_dist, max_features)
        ok, i1, i2, H = self.ransac2d(s1, s2, ransac_dist, 1000, seed)          # CMosaicHarris::Ransac == Ransac2D arithmetic (tests/test_surf.py)
        nin = len(i1)
        return (nin if nin > min_inliers else 0), i1, i2, H, len(s1)


class Ref:
    """The reference's own code (oracle/_ref)."""

    def __init__(self, path):
        self.L = C.CDLL(path)

    def inverse_matrix(self, a, eps):
        a = np.ascontiguousarray(a, np.float32)
        n = a.shape[0]
        out = np.zeros((n, n), np.float32)
        rc = self.L.ref_inverse_matrix(_p(a), n, _p(out), C.c_float(eps))
        return rc, out

    def solve_homography(self, p1, p2):
        a = np.ascontiguousarray(np.stack([p1["x"], p1["y"]], 1), np.float32)
        b = np.ascontiguousarray(np.stack([p2["x"], p2["y"]], 1), np.float32)
        H = np.zeros(9, np.float32)
        rc = self.L.ref_solve_homography(_p(a), _p(b), len(a), _p(H))
        return rc, H

    def nlls(self, p1, p2, H0):
        a = np.ascontiguousarray(np.stack([p1["x"], p1["y"]], 1), np.float32)
        b = np.ascontiguousarray(np.stack([p2["x"], p2["y"]], 1), np.float32)
        H = np.zeros(9, np.float32)
        H0 = np.ascontiguousarray(H0, np.float32)
        rc = self.L.ref_nlls(_p(a), _p(b), len(a), _p(H0), _p(H))
        return rc, H

    def ransac2d(self, p1, p2, dist, sample_times, seed):
        n = len(p1)
        i1 = np.zeros(max(n, 1), SFPOINT)
        i2 = np.zeros(max(n, 1), SFPOINT)
        nin = C.c_int(0)
        H = np.zeros(9, np.float32)
        ok = self.L.ref_ransac2d(_p(p1), _p(p2), n, C.c_float(dist), sample_times, C.c_uint(seed),
                                 _p(i1), _p(i2), C.byref(nin), _p(H))
        return ok, i1[:nin.value].copy(), i2[:nin.value].copy(), H

    def select(self, matches, kp1, kp2, nMatch, w, h, gx=3, gy=3):
        matches = np.ascontiguousarray(matches, np.int32)
        kp1 = np.ascontiguousarray(kp1, np.float32)
        kp2 = np.ascontiguousarray(kp2, np.float32)
        n = len(matches)
        o1 = np.zeros(max(n, 1), SFPOINT)
        o2 = np.zeros(max(n, 1), SFPOINT)
        no = C.c_int(0)
        self.L.ref_select_match_pairs(_p(matches), n, _p(kp1), len(kp1), _p(kp2), len(kp2), nMatch, w, h, gx, gy,
                                      _p(o1), _p(o2), C.byref(no))
        return o1[:no.value].copy(), o2[:no.value].copy()

    def image_projection_transform(self, img, h9):
        img = np.ascontiguousarray(img)
        hh, ws = img.shape[0], img.strides[0]
        ch = img.shape[2] if img.ndim == 3 else 1
        w = img.shape[1]
        h9 = np.ascontiguousarray(h9, np.float32).copy()
        dst = C.c_void_p()
        dw, dh, dws = C.c_int(), C.c_int(), C.c_int()
        rc = self.L.ref_image_projection_transform(_p(img), w, hh, ws, ch, _p(h9), C.byref(dst), C.byref(dw), C.byref(dh), C.byref(dws))
        if rc != 0 or not dst.value:
            return rc, None
        buf = np.ctypeslib.as_array(C.cast(dst, u8p), shape=(dh.value, dws.value)).copy()
        self.L.ref_free_u8(dst)
        return rc, (buf, dw.value, dh.value, dws.value)

    def mosaic_images_refined(self, imgs, h9s):
        n = len(imgs)
        imgs = [np.ascontiguousarray(i) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        cw, ch, cws = C.c_int(), C.c_int(), C.c_int()
        self.L.ref_mosaic_images_refined(ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), None, C.byref(cw), C.byref(ch), C.byref(cws))
        canvas = np.zeros((ch.value, cws.value), np.uint8)
        rc = self.L.ref_mosaic_images_refined(ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(canvas), C.byref(cw), C.byref(ch), C.byref(cws))
        return rc, (canvas, cw.value, ch.value, cws.value)


    def resample_by_overlap(self, w, h, h9s, overlapT=0.7):
        """the reference's ResampleByOverlap (MosaicImage.cpp:2069-2201)"""
        w = np.ascontiguousarray(w, np.int32); h = np.ascontiguousarray(h, np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep = np.zeros(len(w), np.int32)
        self.L.ref_resample_by_overlap(_p(w), _p(h), len(w), _p(h9s), C.c_float(overlapT), _p(keep))
        return keep.astype(np.uint8)

    def chips_and_masks(self, imgs, h9s, keep=None, find_masks=True, res_scale=1.0):
        """the reference's LaplacianPyramidBlending warp stage (MosaicImage.cpp:2216-2460) + FindMasksByDistMap (:1761-1881);
        keep=None runs the reference's own ResampleByOverlap(0.7)"""
        n = len(imgs)
        imgs = [np.ascontiguousarray(i) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        cw, ch = C.c_int(), C.c_int()
        nv = self.L.ref_lpb_run(ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), C.c_float(res_scale),
                                _p(keep_a) if keep_a is not None else None, int(bool(find_masks)), C.byref(cw), C.byref(ch))
        chips = np.zeros(nv, CHIPINFO)
        cimgs, masks = [], []
        for k in range(nv):
            g = np.zeros(5, np.int32); q = np.zeros(8, np.float32)
            self.L.ref_lpb_chip(k, _p(g), _p(q), None, None)
            cws, mws = (int(g[2]) * 3 + 3) & ~3, (int(g[2]) + 3) & ~3
            chip = np.zeros((int(g[3]), cws), np.uint8); mask = np.zeros((int(g[3]), mws), np.uint8)
            self.L.ref_lpb_chip(k, _p(g), _p(q), _p(chip), _p(mask))
            chips[k]["x0"], chips[k]["y0"], chips[k]["w"], chips[k]["h"], chips[k]["img"] = g
            chips[k]["quad"] = q
            cimgs.append(chip); masks.append(mask)
        return dict(cw=cw.value, ch=ch.value, chips=chips, chip_imgs=cimgs, masks=masks)


_ORC = None
_ORC_FAST = None
_REF = None


def cpu_has_v3():
    try:
        flags = open("/proc/cpuinfo").read()
        return all(f in flags for f in (" avx2", " fma", " bmi2"))
    except OSError:
        return False


def load_oracle_fast():
    """liboracle_v3.so where the host CPU has AVX2+FMA (same sources, same arithmetic: contraction stays off and fmaf()
    is fused by definition either way), else the portable build.  For the config-sized tests (12 MP frames)."""
    global _ORC_FAST
    if _ORC_FAST is None:
        build_oracle()
        so = os.path.join(ORC_DIR, "liboracle_v3.so")
        _ORC_FAST = Oracle(so) if cpu_has_v3() and os.path.exists(so) else load_oracle()
    return _ORC_FAST


def parallel_map(fn, items, threads=None):
    """runs fn over items on host threads (ctypes releases the GIL; the oracle's C code is reentrant)"""
    import concurrent.futures as cf
    threads = threads or max(1, min(8, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(fn, items))


def load_oracle():
    global _ORC
    if _ORC is None:
        _ORC = Oracle(build_oracle())
    return _ORC


def load_ref():
    global _REF
    if _REF is None:
        so = os.path.join(ORC_DIR, "_ref", "libref_oracle.so")
        if not os.path.exists(so):
            if os.path.isdir("/root/reference"):
                subprocess.check_call(["bash", os.path.join(ORC_DIR, "ref", "build_ref.sh")])
            else:
                return None
        _REF = Ref(so)
    return _REF
