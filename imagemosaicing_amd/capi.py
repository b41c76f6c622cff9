"""ctypes binding of libmi355mosaic.so (include/mi355_mosaic.h).  No torch types cross this boundary:
device buffers are passed as integer addresses (e.g. tensor.data_ptr())."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SFPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("id", "<i4")])
KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
MATCHPAIR = np.dtype([("ax", "<f4"), ("ay", "<f4"), ("aid", "<i4"), ("ai", "<i4"), ("af", "<i4"),
                      ("bx", "<f4"), ("by", "<f4"), ("bid", "<i4"), ("bi", "<i4"), ("bf", "<i4")])
PAIR_RESULT = np.dtype([("i", "<i4"), ("j", "<i4"), ("n_in", "<i4"), ("n_selected", "<i4"), ("ok", "<i4"),
                        ("accepted", "<i4"), ("H", "<f4", (9,)), ("_pad", "<i4"),
                        ("a", SFPOINT, (400,)), ("b", SFPOINT, (400,))])
CHIPINFO = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("w", "<i4"), ("h", "<i4"), ("img", "<i4"),
                     ("sx", "<f4"), ("sy", "<f4"), ("quad", "<f4", (8,))])
IMAGE_TRANSFORM = np.dtype([("m", "<f4", (9,)), ("fixed", "<i4")])
PAIR_MOMENTS = np.dtype([("i", "<i4"), ("j", "<i4"), ("n_in", "<i4"), ("_pad", "<i4"), ("aa", "<f8", (6,)), ("ab", "<f8", (9,)), ("bb", "<f8", (6,))])
FEATURE_HEADER = np.dtype([("img_id", "<i4"), ("n_kp", "<i4"), ("w", "<i4"), ("h", "<i4")])
FEATURE_RECORD_BYTES = 319488
assert SFPOINT.itemsize == 12 and KEYPOINT.itemsize == 28 and MATCHPAIR.itemsize == 40 and PAIR_RESULT.itemsize == 9664 and PAIR_MOMENTS.itemsize == 184


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("n_octave_layers", C.c_int32), ("contrast_threshold", C.c_float),
                ("edge_threshold", C.c_float), ("sigma", C.c_float), ("max_selected", C.c_int32),
                ("select_fraction", C.c_double), ("grid_x", C.c_int32), ("grid_y", C.c_int32), ("min_inliers", C.c_int32),
                ("ransac_dist", C.c_float), ("sample_times", C.c_int32), ("pair_window", C.c_int32), ("ratio", C.c_float)]


class Mi355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi355 error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    # MI355_LIB: a kernel A/B build of the same library (scratch/build_variant.sh); measurement only
    return os.environ.get("MI355_LIB") or os.path.join(_HERE, "libmi355mosaic.so")


def load_library():
    """Loads the in-tree HIP library.  Fails loudly when it has not been built (python -m imagemosaicing_amd.build)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("libmi355mosaic.so is not built: run `python -m imagemosaicing_amd.build` "
                              "(there is no CPU fallback for the HIP path)")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.
        # When torch is present (bench.py, tests: device memory + torch.distributed), import it FIRST so that this
        # library's DT_NEEDED libamdhip64.so.7 resolves to the runtime torch already loaded instead of a second copy
        # from /opt/rocm (two runtimes in one process cannot both own the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(p)
        L.mi355_last_error.restype = C.c_char_p
        L.mi355_last_error.argtypes = [C.c_void_p]
        L.mi355_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.mi355_destroy.argtypes = [C.c_void_p]
        L.mi355_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def default_params():
    p = Params()
    load_library().mi355_default_params(C.byref(p))
    return p


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _img_geom(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = img.shape[2] if img.ndim == 3 else 1
    return img, w, h, img.strides[0], ch


def _copy_out(ptr, nbytes, dtype):
    """Copies nbytes from a library-owned host buffer into a fresh numpy array of dtype."""
    if nbytes == 0:
        return np.zeros(0, dtype)
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy()
    return raw.view(dtype)


def _view_out(ptr, nbytes, dtype, copy):
    """A library-owned host buffer as a numpy array of dtype: a private copy, or (copy=False) a view valid as long as the library says."""
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
    return (raw.copy() if copy else raw).view(dtype)


class Context:
    """One GPU context (mi355_create).  Methods are named after the reference functions they replace."""

    def __init__(self, device=0, params=None):
        self.L = load_library()
        self._h = C.c_void_p()
        rc = self.L.mi355_create(C.byref(self._h), C.byref(params) if params is not None else None, int(device))
        if rc != 0:
            raise Mi355Error(rc, (self.L.mi355_last_error(None) or b"").decode())
        self.params = params if params is not None else default_params()
        self.device = int(device)

    def close(self):
        if self._h:
            self.L.mi355_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise Mi355Error(rc, (self.L.mi355_last_error(self._h) or b"").decode())
        return rc

    # ---- plumbing ----------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._chk(self.L.mi355_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._chk(self.L.mi355_synchronize(self._h))

    def set_option(self, name, value):
        self._chk(self.L.mi355_set_option(self._h, name.encode(), int(value)))

    def profile_enable(self, on=True):
        self._chk(self.L.mi355_profile_enable(self._h, int(bool(on))))

    def profile_only(self, cls=None):
        self._chk(self.L.mi355_profile_only(self._h, (cls or "").encode()))

    def profile_reset(self):
        self._chk(self.L.mi355_profile_reset(self._h))

    def profile_get(self, cls):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        self._chk(self.L.mi355_profile_get(self._h, cls.encode(), C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    # ---- features (SiftExtraction_Thread, MosaicWithoutPos.cpp:4832-4887) ------------------------
    def SiftExtract(self, img_id, bgr, max_kp=None):
        img, w, h, ws, ch = _img_geom(bgr)
        assert ch == 3
        max_kp = max_kp or max(int(self.params.nfeatures), 2048)      # nfeatures + ties with the last one (retainBest), up to the record's 2048
        kp = np.zeros(max_kp, KEYPOINT)
        desc = np.zeros((max_kp, 128), np.float32)
        n = C.c_int(0)
        self._chk(self.L.mi355_sift_extract(self._h, int(img_id), _p(img), w, h, ws, _p(kp), _p(desc), max_kp, C.byref(n)))
        k = min(n.value, max_kp)
        return kp[:k].copy(), desc[:k].copy()

    def SiftExtractHost(self, img_id, bgr):
        """Host frame in, nothing back: the frame is copied into the library's staging ring and joins a batch like a device
        frame (mi355_sift_extract with kp = desc = n_kp = NULL).  The array may be reused as soon as the call returns."""
        img, w, h, ws, ch = _img_geom(bgr)
        assert ch == 3
        self._chk(self.L.mi355_sift_extract(self._h, int(img_id), _p(img), w, h, ws, None, None, 0, None))

    def SiftExtractDev(self, img_id, d_bgr, w, h, ws, want_count=False):
        """Asynchronous unless want_count: the frame is enqueued on one of the library's SIFT streams and the call
        returns; the keypoint count is adopted at the next MatchPairs / GetFeatures / synchronize."""
        if not want_count:
            self._chk(self.L.mi355_sift_extract_dev(self._h, int(img_id), C.c_void_p(int(d_bgr)), int(w), int(h), int(ws), None))
            return None
        n = C.c_int(0)
        self._chk(self.L.mi355_sift_extract_dev(self._h, int(img_id), C.c_void_p(int(d_bgr)), int(w), int(h), int(ws), C.byref(n)))
        return n.value

    def GetFeatures(self, img_id, max_kp=4096):
        kp = np.zeros(max_kp, KEYPOINT)
        desc = np.zeros((max_kp, 128), np.float32)
        n = C.c_int(0)
        self._chk(self.L.mi355_get_features(self._h, int(img_id), _p(kp), _p(desc), max_kp, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def SetFeatures(self, img_id, kp, desc, w, h):
        kp = np.ascontiguousarray(kp, KEYPOINT)
        desc = np.ascontiguousarray(desc, np.float32)
        self._chk(self.L.mi355_set_features(self._h, int(img_id), _p(kp), _p(desc), len(kp), int(w), int(h)))

    def DropFeatures(self, img_id=-1):
        self._chk(self.L.mi355_drop_features(self._h, int(img_id)))

    # ---- match (GetMatchedPairsOneToAllSIFTThread j-loop, MosaicWithoutPos.cpp:5084-5232) ----------
    def MatchPairs(self, pairs, ransac_dist=2.5, seed=1):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        out = np.zeros(len(pairs), PAIR_RESULT)
        self._chk(self.L.mi355_match_pairs(self._h, _p(pairs), len(pairs), C.c_float(ransac_dist), C.c_uint32(seed), _p(out)))
        return out

    def MatchPairsDev(self, pairs, d_out, ransac_dist=2.5, seed=1):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self._chk(self.L.mi355_match_pairs_dev(self._h, _p(pairs), len(pairs), C.c_float(ransac_dist), C.c_uint32(seed), C.c_void_p(int(d_out))))

    def BFMatch(self, img_i, img_j, sorted_=True, max_matches=2048):
        m = np.zeros(max_matches, DMATCH)
        d2 = np.zeros(max_matches, np.int32)
        s2 = np.zeros(max_matches, np.int32)
        n = C.c_int(0)
        self._chk(self.L.mi355_bf_match(self._h, int(img_i), int(img_j), int(bool(sorted_)), _p(m), _p(d2), _p(s2), max_matches, C.byref(n)))
        k = min(n.value, max_matches)
        return m[:k].copy(), d2[:k].copy(), s2[:k].copy()

    # ---- stand-alone reference functions ------------------------------------------------------------
    def SelectMatchPairs(self, matches, kp1_xy, kp2_xy, nMatch, width, height, gridX=3, gridY=3):
        """MosaicWithoutPos.cpp:4977-5028.  matches: DMATCH array (already sorted) or (n,2) int array."""
        matches = np.asarray(matches)
        if matches.dtype != DMATCH:
            mm = np.zeros(len(matches), DMATCH)
            mm["queryIdx"] = matches[:, 0]
            mm["trainIdx"] = matches[:, 1]
            matches = mm
        matches = np.ascontiguousarray(matches)
        kp1 = np.ascontiguousarray(kp1_xy, np.float32)
        kp2 = np.ascontiguousarray(kp2_xy, np.float32)
        v1 = np.zeros(400, SFPOINT)
        v2 = np.zeros(400, SFPOINT)
        n = C.c_int(0)
        self._chk(self.L.mi355_select_grid(self._h, _p(matches), len(matches), _p(kp1), len(kp1), _p(kp2), len(kp2), int(nMatch),
                                           int(width), int(height), int(gridX), int(gridY), _p(v1), _p(v2), C.byref(n)))
        return v1[:n.value].copy(), v2[:n.value].copy()

    def Ransac2D(self, p1, p2, fRansacDist=1.0, sampleTimes=1000, seed=1):
        """mosaicimage.h:1729-2035.  Returns (ok, inliers1, inliers2, H[9])."""
        p1 = np.ascontiguousarray(p1, SFPOINT)
        p2 = np.ascontiguousarray(p2, SFPOINT)
        n = len(p1)
        i1 = np.zeros(max(400, n), SFPOINT)
        i2 = np.zeros(max(400, n), SFPOINT)
        nin = C.c_int(0)
        H = np.zeros(9, np.float32)
        ok = self._chk(self.L.mi355_ransac2d(self._h, _p(p1), _p(p2), n, C.c_float(fRansacDist), int(sampleTimes), C.c_uint32(seed),
                                             _p(i1), _p(i2), C.byref(nin), _p(H)))
        return ok, i1[:nin.value].copy(), i2[:nin.value].copy(), H

    def ImageProjectionTransform(self, img, h9):
        """MosaicImage.cpp:1613-1758.  Returns (rows x widthStep u8 buffer, width, height, widthStep)."""
        img, w, h, ws, ch = _img_geom(img)
        h9 = np.ascontiguousarray(h9, np.float32)
        dst = C.c_void_p()
        dw, dh, dws = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mi355_warp_image(self._h, _p(img), w, h, ws, ch, _p(h9), C.byref(dst), C.byref(dw), C.byref(dh), C.byref(dws)))
        buf = _copy_out(dst, dh.value * dws.value, np.uint8).reshape(dh.value, dws.value)
        self.L.mi355_free(dst)
        return buf, dw.value, dh.value, dws.value

    def MosaicImagesRefined(self, imgs, h9s, want_pixels=True):
        """CMosaicByPose::MosaicImagesRefined (float), MosaicWithoutPos.cpp:2194-2352.  want_pixels=False: the canvas the library returns is
        released without the numpy copy (bench.py times the C call, not this binding)."""
        n = len(imgs)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        canvas = C.c_void_p()
        cw, ch, cws = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mi355_mosaic_refined(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), C.byref(canvas), C.byref(cw), C.byref(ch), C.byref(cws)))
        buf = _copy_out(canvas, ch.value * cws.value, np.uint8).reshape(ch.value, cws.value) if want_pixels else None
        self.L.mi355_free(canvas)
        return buf, cw.value, ch.value, cws.value

    def MosaicImagesRefinedDev(self, d_imgs, w, h, ws, h9s, d_canvas, cw, ch, cws, row0=0, rows=-1):
        n = len(d_imgs)
        ptrs = (C.c_void_p * n)(*[int(p) for p in d_imgs])
        w = np.ascontiguousarray(w, np.int32)
        h = np.ascontiguousarray(h, np.int32)
        ws = np.ascontiguousarray(ws, np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        self._chk(self.L.mi355_mosaic_refined_dev(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), C.c_void_p(int(d_canvas)),
                                                  int(cw), int(ch), int(cws), int(row0), int(rows if rows >= 0 else ch)))

    # ---- SURF variant (GetMatchedPairsOneToAllSurf, MosaicWithoutPos.cpp:5300-5533) ---------------------
    def SurfExtract(self, img_id, bgr, hessian=50.0, max_kp=4096):
        img, w, h, ws, ch = _img_geom(bgr)
        assert ch == 3
        kp = np.zeros(max_kp, KEYPOINT)
        desc = np.zeros((max_kp, 128), np.float32)
        n = C.c_int(0)
        self._chk(self.L.mi355_surf_extract(self._h, int(img_id), _p(img), w, h, ws, C.c_float(hessian), int(max_kp), _p(kp), _p(desc), C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def SurfMatchPairs(self, pairs, ransac_dist=2.5, seed=1, match_dist=0.5, max_features=200, min_inliers=18):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        out = np.zeros(len(pairs), PAIR_RESULT)
        self._chk(self.L.mi355_surf_match_pairs(self._h, _p(pairs), len(pairs), C.c_float(ransac_dist), C.c_uint32(seed), C.c_float(match_dist),
                                                int(max_features), int(min_inliers), _p(out)))
        return out

    # ---- multi-GPU exchanges -------------------------------------------------------------
    def PackFeaturesDev(self, img_ids, d_payload):
        """resident features of img_ids -> fixed-size records at d_payload (device, len x FEATURE_RECORD_BYTES); returns the headers"""
        ids = np.ascontiguousarray(img_ids, np.int32)
        hdr = np.zeros(len(ids), FEATURE_HEADER)
        self._chk(self.L.mi355_pack_features_dev(self._h, _p(ids), len(ids), _p(hdr), C.c_void_p(int(d_payload))))
        return hdr

    def InstallFeaturesDev(self, hdr, d_payload):
        hdr = np.ascontiguousarray(hdr, FEATURE_HEADER)
        self._chk(self.L.mi355_install_features_dev(self._h, _p(hdr), C.c_void_p(int(d_payload)), len(hdr)))

    def CompactAcceptedDev(self, d_in, n, d_out):
        k = C.c_int(0)
        self._chk(self.L.mi355_compact_accepted_dev(self._h, C.c_void_p(int(d_in)), int(n), C.c_void_p(int(d_out)), C.byref(k)))
        return k.value

    def last_sift_counters(self):
        """SIFT stage populations of the last extracted frame: DoG extrema, refined points, oriented keypoints, kept, overflow flag"""
        out = (C.c_int32 * 8)()
        self._chk(self.L.mi355_last_sift_counters(self._h, out))
        return list(out)

    def CommInit(self, id128, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self._chk(self.L.mi355_comm_init(self._h, buf, int(rank), int(world)))

    def CommDestroy(self):
        self._chk(self.L.mi355_comm_destroy(self._h))

    def CommInfo(self):
        """(rank, n_ranks) as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount)"""
        r, n = C.c_int(0), C.c_int(0)
        self._chk(self.L.mi355_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def AllGatherFeatures(self, img_ids, n_max_per_rank):
        ids = np.ascontiguousarray(img_ids, np.int32)
        self._chk(self.L.mi355_allgather_features(self._h, _p(ids), len(ids), int(n_max_per_rank)))

    def AllGatherResults(self, d_local, n_local, accepted_only=True, root=-1, copy=True, wait=True):
        """mi355_allgather_results.  root < 0: every rank receives all ranks' records; root >= 0: only that rank (the others send and
        get an empty array).  The library hands out a view of pinned memory it owns, valid until the next call: copy=True (default)
        returns a private numpy copy, copy=False the view itself (bench.py: a 1.1 GB numpy copy per step would be the measurement).
        wait=False (root >= 0, copy=False): MI355_GATHER_NO_WAIT -- the view is complete after the next synchronize()."""
        assert wait or (root >= 0 and not copy)
        ptr, n = C.c_void_p(), C.c_int(0)
        flags = (1 if accepted_only else 0) | (0 if wait else 2)
        self._chk(self.L.mi355_allgather_results(self._h, C.c_void_p(int(d_local)), int(n_local), flags, int(root), C.byref(ptr), C.byref(n)))
        if not ptr.value or n.value == 0:
            return np.zeros(0, PAIR_RESULT)
        return _view_out(ptr, n.value * PAIR_RESULT.itemsize, PAIR_RESULT, copy)

    def PairMomentsDev(self, d_results, n, d_out):
        """one PAIR_MOMENTS record per pair record, device to device (ctx stream)"""
        self._chk(self.L.mi355_pair_moments_dev(self._h, C.c_void_p(int(d_results)), int(n), C.c_void_p(int(d_out))))

    def AllGatherMoments(self, d_local, n_local, copy=True):
        """this rank's accepted pairs -> their second moments -> every rank's host (rank-major PAIR_MOMENTS array; pinned memory of the
        library, see AllGatherResults for `copy`)"""
        ptr, n = C.c_void_p(), C.c_int(0)
        self._chk(self.L.mi355_allgather_moments(self._h, C.c_void_p(int(d_local)), int(n_local), C.byref(ptr), C.byref(n)))
        if not ptr.value or n.value == 0:
            return np.zeros(0, PAIR_MOMENTS)
        return _view_out(ptr, n.value * PAIR_MOMENTS.itemsize, PAIR_MOMENTS, copy)

    def StripeCover(self, w, h, h9s, row0, rows, blended=False, keep=None, band=5, exact=False):
        """mi355_mosaic_stripe_cover: need[k] = 1 when rendering canvas rows [row0, row0 + rows) reads frame k.  exact (refined canvas only): the
        frames that give at least one pixel its sample (a device pass), instead of every frame whose box meets the rows"""
        n = len(w)
        w = np.ascontiguousarray(w, np.int32); h = np.ascontiguousarray(h, np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        need = np.zeros(n, np.uint8)
        mode = 1 if blended else (2 if exact else 0)
        self._chk(self.L.mi355_mosaic_stripe_cover(self._h, mode, _p(w), _p(h), n, _p(h9s), _p(keep_a), int(band), int(row0), int(rows), _p(need)))
        return need

    def ExchangeFrames(self, d_frames, h, ws, need, owner=None, own_through_rccl=False):
        """mi355_exchange_frames.  d_frames: per frame this rank's device pointer (0 / None where it does not hold the frame); need: [G, n]
        uint8 table (the same on every rank), or [n]: this rank's own row (the rows are all-gathered inside the call).  Returns (pointers for
        the stripe calls -- 0 where the rank's stripe does not read the frame --, bytes received, bytes sent)."""
        n = len(d_frames)
        ptrs = (C.c_void_p * max(n, 1))(*[int(p) if p else None for p in d_frames])
        h = np.ascontiguousarray(h, np.int32); ws = np.ascontiguousarray(ws, np.int32)
        need = np.ascontiguousarray(need, np.uint8)
        local = need.ndim == 1
        assert need.shape[-1] == n
        own_a = None if owner is None else np.ascontiguousarray(owner, np.int32)
        out = (C.c_void_p * max(n, 1))()
        br, bs = C.c_uint64(0), C.c_uint64(0)
        flags = (1 if own_through_rccl else 0) | (2 if local else 0)
        self._chk(self.L.mi355_exchange_frames(self._h, ptrs, _p(h), _p(ws), n, _p(own_a), _p(need), flags, out, C.byref(br), C.byref(bs)))
        return [int(out[k] or 0) for k in range(n)], int(br.value), int(bs.value)

    def SynthFrameDev(self, d_dst, w, h, ws, A6, seed, frame_seed, gain=1.0, noise=2.0):
        A6 = np.ascontiguousarray(A6, np.float32)
        self._chk(self.L.mi355_synth_frame_dev(self._h, C.c_void_p(int(d_dst)), int(w), int(h), int(ws), _p(A6), C.c_uint32(seed),
                                               C.c_uint32(frame_seed), C.c_float(gain), C.c_float(noise)))

    def ChipsAndMasks(self, imgs, h9s, keep=None, find_masks=True):
        """LaplacianPyramidBlending warp stage + FindMasksByDistMap (MosaicImage.cpp:2233-2460, 1761-1881)."""
        n = len(imgs)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        nch = C.c_int(0)
        chips = C.c_void_p()
        cimgs = C.POINTER(C.c_void_p)()
        masks = C.POINTER(C.c_void_p)()
        cw, ch = C.c_int(), C.c_int()
        self._chk(self.L.mi355_chips_and_masks(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(keep_a), int(bool(find_masks)),
                                               C.byref(nch), C.byref(chips), C.byref(cimgs), C.byref(masks), C.byref(cw), C.byref(ch)))
        nv = nch.value
        info = _copy_out(chips, nv * CHIPINFO.itemsize, CHIPINFO)
        out_c, out_m = [], []
        for v in range(nv):
            cwv, chv = int(info[v]["w"]), int(info[v]["h"])
            cws_, mws = (cwv * 3 + 3) & ~3, (cwv + 3) & ~3
            out_c.append(_copy_out(cimgs[v], chv * cws_, np.uint8).reshape(chv, cws_))
            out_m.append(_copy_out(masks[v], chv * mws, np.uint8).reshape(chv, mws))
            self.L.mi355_free(C.c_void_p(cimgs[v]))
            self.L.mi355_free(C.c_void_p(masks[v]))
        self.L.mi355_free(chips)
        self.L.mi355_free(C.cast(cimgs, C.c_void_p))
        self.L.mi355_free(C.cast(masks, C.c_void_p))
        return dict(cw=cw.value, ch=ch.value, chips=info, chip_imgs=out_c, masks=out_m)

    def MultiBandBlend(self, chips, chip_imgs, masks, cw, ch, band=5):
        """detail::MultiBandBlender(false, band) over the chips of ChipsAndMasks (MosaicImage.cpp:2296-2299, 2451-2486)."""
        n = len(chip_imgs)
        ci = [np.ascontiguousarray(c, np.uint8) for c in chip_imgs]
        mi = [np.ascontiguousarray(m, np.uint8) for m in masks]
        cp = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in ci])
        mp = (C.c_void_p * max(n, 1))(*[m.ctypes.data for m in mi])
        info = np.ascontiguousarray(chips, CHIPINFO)
        out = C.c_void_p()
        ow, oh, ows = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mi355_multiband_blend(self._h, cp, mp, _p(info), n, int(cw), int(ch), int(band), C.byref(out), C.byref(ow), C.byref(oh), C.byref(ows)))
        buf = _copy_out(out, ows.value * oh.value, np.uint8).reshape(oh.value, ows.value)
        self.L.mi355_free(out)
        return buf, ow.value, oh.value, ows.value

    def MosaicBlended(self, imgs, h9s, keep=None, band=5):
        """LaplacianPyramidBlending in one call (MosaicImage.cpp:2205-2510): chips, masks and blend stay on the device."""
        n = len(imgs)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        w = np.array([i.shape[1] for i in imgs], np.int32)
        h = np.array([i.shape[0] for i in imgs], np.int32)
        ws = np.array([i.strides[0] for i in imgs], np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        out = C.c_void_p()
        ow, oh, ows = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mi355_mosaic_blended(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(keep_a), int(band), C.byref(out), C.byref(ow), C.byref(oh), C.byref(ows)))
        buf = _copy_out(out, ows.value * oh.value, np.uint8).reshape(oh.value, ows.value)
        self.L.mi355_free(out)
        return buf, ow.value, oh.value, ows.value


    def MosaicBlendedDev(self, d_ptrs, w, h, ws, h9s, keep=None, band=5, row0=0, rows=-1):
        """LaplacianPyramidBlending with the survey resident in HBM: device frames in, device canvas out (a torch uint8 tensor
        [ch, cws]); chips, masks and the blender's pyramids stay in the ctx's buffers.
        Ordering: the library writes the canvas on the ctx's stream, the tensor comes from torch's allocator (torch's current stream).  The
        wrapper synchronises torch's stream before the call (a recycled block may still be in use by pending torch work) and the ctx's
        stream after it: the tensor it returns is complete and safe to use on any stream."""
        import torch
        n = len(d_ptrs)
        ptrs = (C.c_void_p * n)(*[int(p) for p in d_ptrs])
        w = np.ascontiguousarray(w, np.int32); h = np.ascontiguousarray(h, np.int32); ws = np.ascontiguousarray(ws, np.int32)
        h9s = np.ascontiguousarray(h9s, np.float32)
        keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        cw, ch, cws = blend_layout(w, h, h9s, keep_a)
        if rows >= 0:
            # one stripe of the canvas (a rank's share): the tensor holds the rows row0 .. row0 + rows - 1; cw, ch, cws stay the whole canvas's
            out = torch.empty((rows, cws), dtype=torch.uint8, device=torch.device("cuda", self.device))
            torch.cuda.current_stream(out.device).synchronize()
            self._chk(self.L.mi355_mosaic_blended_rows_dev(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(keep_a), int(band),
                                                           C.c_void_p(out.data_ptr()), cw, ch, cws, int(row0), int(rows)))
            self.synchronize()
            return out, cw, ch, cws
        out = torch.empty((ch, cws), dtype=torch.uint8, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(out.device).synchronize()
        self._chk(self.L.mi355_mosaic_blended_dev(self._h, ptrs, _p(w), _p(h), _p(ws), n, _p(h9s), _p(keep_a), int(band),
                                                  C.c_void_p(out.data_ptr()), cw, ch, cws))
        self.synchronize()
        return out, cw, ch, cws


# ---- host-only helpers (no ctx) ---------------------------------------------------------------------------
def comm_unique_id():
    """128-byte RCCL id for mi355_comm_init (rank 0 creates it, every rank receives it by any transport)"""
    L = load_library()
    buf = (C.c_uint8 * 128)()
    rc = L.mi355_comm_unique_id(buf)
    if rc != 0:
        raise Mi355Error(rc, "comm_unique_id: librccl not usable")
    return bytes(buf)


def blend_layout(w, h, h9s, keep=None):
    """canvas size (cw, ch, cws) of LaplacianPyramidBlending for these transforms (MosaicImage.cpp:2233-2292)"""
    L = load_library()
    w = np.ascontiguousarray(w, np.int32); h = np.ascontiguousarray(h, np.int32)
    h9s = np.ascontiguousarray(h9s, np.float32)
    keep_a = None if keep is None else np.ascontiguousarray(keep, np.uint8)
    cw, ch, cws = C.c_int(), C.c_int(), C.c_int()
    rc = L.mi355_blend_layout(_p(w), _p(h), len(w), _p(h9s), _p(keep_a), C.byref(cw), C.byref(ch), C.byref(cws))
    if rc != 0:
        raise Mi355Error(rc, "blend_layout")
    return cw.value, ch.value, cws.value


def comm_available():
    """True when the library can bind librccl in this process (touches no communicator)"""
    return load_library().mi355_comm_available() == 0


def mosaic_layout(w, h, h9s):
    L = load_library()
    w = np.ascontiguousarray(w, np.int32)
    h = np.ascontiguousarray(h, np.int32)
    h9s = np.ascontiguousarray(h9s, np.float32)
    cw, ch, cws = C.c_int(), C.c_int(), C.c_int()
    dG = np.zeros(2, np.float32)
    rc = L.mi355_mosaic_layout(_p(w), _p(h), len(w), _p(h9s), C.byref(cw), C.byref(ch), C.byref(cws), _p(dG))
    if rc != 0:
        raise Mi355Error(rc, "mosaic_layout")
    return cw.value, ch.value, cws.value, dG


def resample_by_overlap(w, h, h9s, overlapT=0.7):
    """ResampleByOverlap (MosaicImage.cpp:2069-2201): keep[k] = vecAbandonInd[k]"""
    L = load_library()
    w = np.ascontiguousarray(w, np.int32)
    h = np.ascontiguousarray(h, np.int32)
    h9s = np.ascontiguousarray(h9s, np.float32)
    keep = np.zeros(len(w), np.uint8)
    rc = L.mi355_resample_by_overlap(_p(w), _p(h), len(w), _p(h9s), C.c_float(overlapT), _p(keep))
    if rc != 0:
        raise Mi355Error(rc, "resample_by_overlap")
    return keep


def surf_pair_schedule(n_images):
    L = load_library()
    n = C.c_int(0)
    L.mi355_surf_pair_schedule(int(n_images), None, 0, C.byref(n))
    out = np.zeros((max(n.value, 1), 2), np.int32)
    L.mi355_surf_pair_schedule(int(n_images), _p(out), n.value, C.byref(n))
    return out[:n.value]


def pair_schedule(n_images, window, rank=0, world=1):
    L = load_library()
    n = C.c_int(0)
    L.mi355_pair_schedule(int(n_images), int(window), int(rank), int(world), None, 0, C.byref(n))
    out = np.zeros((max(n.value, 1), 2), np.int32)
    rc = L.mi355_pair_schedule(int(n_images), int(window), int(rank), int(world), _p(out), n.value, C.byref(n))
    if rc != 0:
        raise Mi355Error(rc, "pair_schedule")
    return out[:n.value]


def write_match_pairs(path, v):
    v = np.ascontiguousarray(v, MATCHPAIR)
    rc = load_library().mi355_write_match_pairs(path.encode(), _p(v), len(v))
    if rc != 0:
        raise Mi355Error(rc, "write_match_pairs")


def load_match_pairs(path):
    L = load_library()
    ptr, n = C.c_void_p(), C.c_int(0)
    rc = L.mi355_load_match_pairs(path.encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise Mi355Error(rc, "load_match_pairs")
    out = _copy_out(ptr, n.value * 40, MATCHPAIR)
    L.mi355_free(ptr)
    return out


def write_match_pairs_txt(path, v):
    v = np.ascontiguousarray(v, MATCHPAIR)
    rc = load_library().mi355_write_match_pairs_txt(path.encode(), _p(v), len(v))
    if rc != 0:
        raise Mi355Error(rc, "write_match_pairs_txt")


def write_transforms(path, t):
    t = np.ascontiguousarray(t, IMAGE_TRANSFORM)
    rc = load_library().mi355_write_transforms(path.encode(), _p(t), len(t))
    if rc != 0:
        raise Mi355Error(rc, "write_transforms")


def load_transforms(path, tran0=False):
    """ImportTransform's format (count + 9 floats each), or with tran0=True the rows OutTransform writes (tran0.txt)"""
    L = load_library()
    ptr, n = C.c_void_p(), C.c_int(0)
    rc = (L.mi355_load_tran0 if tran0 else L.mi355_load_transforms)(path.encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise Mi355Error(rc, "load_transforms")
    out = _copy_out(ptr, n.value * IMAGE_TRANSFORM.itemsize, IMAGE_TRANSFORM)
    L.mi355_free(ptr)
    return out


def write_keypoints(path, kp):
    kp = np.ascontiguousarray(kp, KEYPOINT)
    rc = load_library().mi355_write_keypoints(path.encode(), _p(kp), len(kp))
    if rc != 0:
        raise Mi355Error(rc, "write_keypoints")


def load_keypoints(path):
    L = load_library()
    ptr, n = C.c_void_p(), C.c_int(0)
    rc = L.mi355_load_keypoints(path.encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise Mi355Error(rc, "load_keypoints")
    out = _copy_out(ptr, n.value * 28, KEYPOINT)
    L.mi355_free(ptr)
    return out


def write_descriptors_xml(path, desc):
    """discriptor_%d.xml (cv::FileStorage << "descriptor" << Mat CV_32F), OpenCV 2.4's XML layout (unpinned: the reference commits no such file)"""
    d = np.ascontiguousarray(desc, np.float32)
    if d.ndim != 2:
        raise ValueError("descriptors: a 2-D array")
    rc = load_library().mi355_write_descriptors_xml(path.encode(), _p(d), int(d.shape[0]), int(d.shape[1]))
    if rc != 0:
        raise Mi355Error(rc, "write_descriptors_xml")


def load_descriptors_xml(path):
    L = load_library()
    ptr, r, c = C.c_void_p(), C.c_int(0), C.c_int(0)
    rc = L.mi355_load_descriptors_xml(path.encode(), C.byref(ptr), C.byref(r), C.byref(c))
    if rc != 0:
        raise Mi355Error(rc, "load_descriptors_xml")
    out = _copy_out(ptr, r.value * c.value * 4, np.float32).reshape(r.value, c.value)
    L.mi355_free(ptr)
    return out


def results_to_match_pairs(results, fixed_flags=None):
    L = load_library()
    results = np.ascontiguousarray(results, PAIR_RESULT)
    ff = None if fixed_flags is None else np.ascontiguousarray(fixed_flags, np.int32)
    ptr, n = C.c_void_p(), C.c_int(0)
    rc = L.mi355_results_to_match_pairs(_p(results), len(results), _p(ff), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise Mi355Error(rc, "results_to_match_pairs")
    out = _copy_out(ptr, n.value * 40, MATCHPAIR)
    L.mi355_free(ptr)
    return out


def global_affine_align(match_pairs, n_images, fixed=None):
    v = np.ascontiguousarray(match_pairs, MATCHPAIR)
    ff = None if fixed is None else np.ascontiguousarray(fixed, np.int32)
    out = np.zeros(n_images, IMAGE_TRANSFORM)
    rc = load_library().mi355_global_affine_align(_p(v), len(v), int(n_images), _p(ff), _p(out))
    if rc != 0:
        raise Mi355Error(rc, "global_affine_align")
    return out


def select_connected_results(results, n_images):
    """Select_Connected_Matched_Images straight from PAIR_RESULT records (accepted pairs with inliers are the edges)"""
    r = np.ascontiguousarray(results, PAIR_RESULT)
    label = np.zeros(n_images, np.int32)
    rc = load_library().mi355_select_connected_results(_p(r), len(r), int(n_images), _p(label))
    if rc != 0:
        raise Mi355Error(rc, "select_connected_results")
    return label


def global_affine_align_results(results, n_images, fixed=None, label=None):
    """global_affine_align straight from PAIR_RESULT records; label: use only the pairs whose two images are labelled"""
    r = np.ascontiguousarray(results, PAIR_RESULT)
    ff = None if fixed is None else np.ascontiguousarray(fixed, np.int32)
    lb = None if label is None else np.ascontiguousarray(label, np.int32)
    out = np.zeros(n_images, IMAGE_TRANSFORM)
    rc = load_library().mi355_global_affine_align_results(_p(r), len(r), int(n_images), _p(ff), _p(lb), _p(out))
    if rc != 0:
        raise Mi355Error(rc, "global_affine_align_results")
    return out


def pair_moments_host(results):
    """the second moments of every record's inlier coordinates, summed on the host (what mi355_pair_moments_dev forms on the device)"""
    r = np.ascontiguousarray(results, PAIR_RESULT)
    out = np.zeros(len(r), PAIR_MOMENTS)
    rc = load_library().mi355_pair_moments_host(_p(r), len(r), _p(out))
    if rc != 0:
        raise Mi355Error(rc, "pair_moments_host")
    return out


def select_connected_moments(moments, n_images):
    m = np.ascontiguousarray(moments, PAIR_MOMENTS)
    label = np.zeros(n_images, np.int32)
    rc = load_library().mi355_select_connected_moments(_p(m), len(m), int(n_images), _p(label))
    if rc != 0:
        raise Mi355Error(rc, "select_connected_moments")
    return label


def global_affine_align_moments(moments, n_images, fixed=None, label=None):
    """the global alignment from the pairs' second moments: the same bits as global_affine_align_results on the records they were formed from"""
    m = np.ascontiguousarray(moments, PAIR_MOMENTS)
    ff = None if fixed is None else np.ascontiguousarray(fixed, np.int32)
    lb = None if label is None else np.ascontiguousarray(label, np.int32)
    out = np.zeros(n_images, IMAGE_TRANSFORM)
    rc = load_library().mi355_global_affine_align_moments(_p(m), len(m), int(n_images), _p(ff), _p(lb), _p(out))
    if rc != 0:
        raise Mi355Error(rc, "global_affine_align_moments")
    return out


def select_connected(match_pairs, n_images):
    v = np.ascontiguousarray(match_pairs, MATCHPAIR)
    label = np.zeros(n_images, np.int32)
    rc = load_library().mi355_select_connected(_p(v), len(v), int(n_images), _p(label))
    if rc != 0:
        raise Mi355Error(rc, "select_connected")
    return label
