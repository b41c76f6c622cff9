"""GPU: SIFT detect+describe (HIP) vs the CPU restatement oracle/oracle_sift.c -- bit for bit: same
keypoints in the same order (every field of the 28-byte cv::KeyPoint record) and identical u8 descriptors.
Parity of this stage is UNPINNED by the reference (OpenCV 2.4.0 arithmetic is not available); the oracle
defines it (see its header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import imagemosaicing_amd as im
    c = im.Context(0)
    yield c
    c.close()


def _check(ctx, oracle, img, tag):
    kp, desc = ctx.SiftExtract(7, img)
    okp, odesc = oracle.sift(img)
    assert len(kp) == len(okp), f"{tag}: {len(kp)} vs {len(okp)} keypoints"
    for f in ("octave", "x", "y", "size", "angle", "response", "class_id"):
        a, b = kp[f], okp[f]
        same = a.view(np.uint32) == b.view(np.uint32) if a.dtype.kind == "f" else a == b
        assert same.all(), f"{tag}: field {f} differs at {np.where(~same)[0][:5]} ({a[~same][:3]} vs {b[~same][:3]})"
    d8 = desc.astype(np.uint8)
    assert np.array_equal(desc, d8.astype(np.float32)), "descriptors are not integer valued"
    assert np.array_equal(d8, odesc), f"{tag}: {int((d8 != odesc).any(1).sum())} descriptors differ"
    return kp, d8


def test_sift_two_overlapping_tiles_c1(ctx, oracle):
    """BASELINE config C1: 2 overlapping 640x480 tiles -> detect, describe, match, H"""
    from tests.synth_frames import strip
    frames, Hs = strip(2, 640, 480, seed=1)
    k0, d0 = _check(ctx, oracle, frames[0], "tile0")
    assert len(k0) > 1500                        # a 640x480 tile holds fewer than nfeatures = 2000 keypoints without the doubled octave
    ctx.SiftExtract(0, frames[0]); ctx.SiftExtract(1, frames[1])
    res = ctx.MatchPairs([(0, 1)], 2.5, 1)[0]
    assert int(res["accepted"]) == 1 and int(res["n_in"]) > 100
    # homography maps frame-1 pixels onto frame-0 pixels: compare with ground truth on the frame centre
    Hgt = np.linalg.inv(Hs[0]) @ Hs[1]
    H = res["H"].astype(np.float64).copy(); H[8] = 1.0; H = H.reshape(3, 3)
    p = np.array([320.0, 240.0, 1.0])
    a, b = H @ p, Hgt @ p
    assert np.hypot(*(a[:2] / a[2] - b[:2] / b[2])) < 1.0
    # whole pair vs the oracle pipeline on the oracle's own features
    k1, d1 = oracle.sift(frames[1])
    ok0, od0 = oracle.sift(frames[0])
    nin, i1, i2, Ho, ns = oracle.match_pair(np.stack([ok0["x"], ok0["y"]], 1), od0, np.stack([k1["x"], k1["y"]], 1), d1, 640, 480, 2.5, 1)
    assert nin == int(res["n_in"]) and ns == int(res["n_selected"])
    assert np.array_equal(res["a"][:nin], i1[:nin]) and np.array_equal(res["H"].view(np.uint32), Ho.view(np.uint32))


def test_sift_ragged_sizes(ctx, oracle):
    from tests.synth_frames import terrain
    for (w, h, seed) in [(333, 257, 3), (129, 200, 4), (64, 64, 5)]:
        _check(ctx, oracle, terrain(w, h, seed=seed), f"{w}x{h}")


def test_sift_flat_image_has_no_keypoints(ctx, oracle):
    img = np.full((120, 160, 3), 128, np.uint8)
    kp, desc = ctx.SiftExtract(3, img)
    okp, _ = oracle.sift(img)
    assert len(kp) == 0 and len(okp) == 0


def test_sift_row_padding_is_ignored(ctx, oracle):
    """widthStep > 3*w (IplImage rows padded to 4 bytes): bytes in the padding must not matter"""
    from tests.synth_frames import terrain
    img = terrain(201, 150, seed=9)
    padded = np.zeros((150, 201 * 3 + 5), np.uint8)
    padded[:, :603] = img.reshape(150, -1)
    padded[:, 603:] = 255
    view = np.lib.stride_tricks.as_strided(padded, shape=(150, 201, 3), strides=(padded.strides[0], 3, 1))
    a = ctx.L  # noqa
    import ctypes as C
    import imagemosaicing_amd as im
    kp = np.zeros(2000, im.KEYPOINT); desc = np.zeros((2000, 128), np.float32); n = C.c_int(0)
    rc = ctx.L.mi355_sift_extract(ctx._h, 5, padded.ctypes.data_as(C.c_void_p), 201, 150, padded.strides[0],
                                  kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 2000, C.byref(n))
    assert rc == 0
    kp2, desc2 = ctx.SiftExtract(6, img)
    assert n.value == len(kp2) and np.array_equal(kp[:n.value], kp2) and np.array_equal(desc[:n.value], desc2)
    del view


def test_sift_streaming_blur_levels(ctx, oracle):
    """levels at least 512 columns wide go through blur16_stream (wave-per-strip streaming Gaussian): partial last strip
    (2200 = 8*256 + 152; 1100 = 4*256 + 76; the octaves below halve them), several row segments with reflected top / bottom rows, an
    exact multiple of 256, a width that is no multiple of 4 (tile kernel on every level)"""
    import imagemosaicing_amd as im
    from tests.synth_frames import terrain
    for (w, h, seed) in [(2200, 1604, 20), (1100, 780, 21), (1024, 768, 22), (1101, 700, 23)]:
        img = terrain(w, h, seed=seed)
        kp, d8 = _check(ctx, oracle, img, f"{w}x{h}")
        assert 2000 <= len(kp) <= 2048                 # nfeatures + ties with the last one (retainBest)
        c0 = im.Context(0)
        c0.set_option("blur_stream", 0)
        kp0, desc0 = c0.SiftExtract(7, img)
        c0.close()
        assert np.array_equal(kp0.view(np.uint8), kp.view(np.uint8)) and np.array_equal(desc0.astype(np.uint8), d8)


def test_sift_batched_frames_equal_single(ctx, oracle):
    """frames enqueued asynchronously are processed as batches (small octaves and keypoint stages of all frames in one
    launch each, several work areas in flight): same features as one-at-a-time extraction, for ragged batch sizes, a
    frame-size change inside a stream of frames, and a frame whose big octaves take the streaming kernels"""
    import torch
    import imagemosaicing_amd as im
    from tests.synth_frames import terrain
    sizes = [(320, 240)] * 11 + [(200, 160)] * 3 + [(333, 257)] * 3 + [(1100, 780)] * 3
    imgs = [terrain(w, h, seed=40 + k) for k, (w, h) in enumerate(sizes)]
    ref = []
    for k, img in enumerate(imgs):
        ref.append(ctx.SiftExtract(900 + k, img))          # synchronous: batch of one
    okp, odesc = oracle.sift(imgs[5])
    assert np.array_equal(ref[5][0]["response"].view(np.uint32), okp["response"].view(np.uint32)) and np.array_equal(ref[5][1].astype(np.uint8), odesc)
    for batch, slots in [(8, 3), (4, 1), (3, 2)]:
        c = im.Context(0)
        c.set_option("sift_batch", batch); c.set_option("sift_slots", slots)
        dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        torch.cuda.synchronize()
        for k, d in enumerate(dev):
            c.SiftExtractDev(k, d.data_ptr(), sizes[k][0], sizes[k][1], sizes[k][0] * 3)
        for k in range(len(imgs)):
            kp, desc = c.GetFeatures(k)
            assert np.array_equal(kp.view(np.uint8), ref[k][0].view(np.uint8)), f"batch {batch}: frame {k} keypoints differ"
            assert np.array_equal(desc, ref[k][1]), f"batch {batch}: frame {k} descriptors differ"
        c.close()


def test_sift_streamed_extrema(ctx, oracle):
    """extrema_stream (wave-per-strip DoG extrema with the 3x3x3 neighbourhood taken from registers) against the tiled
    kernel and the oracle: thresholds lowered so that 1100x780 frames take it; includes a flat frame, where every pixel
    is a (tied) extremum and the per-wave candidate list overflows row after row"""
    import torch
    import imagemosaicing_amd as im
    from tests.synth_frames import terrain
    imgs = [terrain(1100, 780, seed=60 + k) for k in range(3)] + [np.full((780, 1100, 3), 77, np.uint8)]
    ref = [ctx.SiftExtract(800 + k, img) for k, img in enumerate(imgs)]      # tiled extrema (one frame per batch)
    okp, odesc = oracle.sift(imgs[0])
    assert np.array_equal(ref[0][0].view(np.uint8), okp.view(np.uint8)) and np.array_equal(ref[0][1].astype(np.uint8), odesc)
    assert len(ref[3][0]) == 0
    c = im.Context(0)
    c.set_option("xstream_min_w", 1000); c.set_option("xstream_min_frames", 1); c.set_option("sift_batch", 4)
    dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    torch.cuda.synchronize()
    for k, d in enumerate(dev):
        c.SiftExtractDev(k, d.data_ptr(), 1100, 780, 3300)
    for k in range(len(imgs)):
        kp, desc = c.GetFeatures(k)
        assert np.array_equal(kp.view(np.uint8), ref[k][0].view(np.uint8)), f"frame {k} keypoints differ"
        assert np.array_equal(desc, ref[k][1]), f"frame {k} descriptors differ"
    c.close()


def test_sift_host_frames_deferred(ctx):
    """mi355_sift_extract with nothing asked back: host frames go through the staging ring and join batches; the source
    array is overwritten right after each call; more frames than the ring holds"""
    import imagemosaicing_amd as im
    from tests.synth_frames import terrain
    imgs = [terrain(320, 240, seed=70 + k) for k in range(21)]
    ref = [ctx.SiftExtract(700 + k, img) for k, img in enumerate(imgs)]
    c = im.Context(0)
    c.set_option("sift_batch", 4); c.set_option("sift_slots", 2)          # ring of 12 frames
    buf = np.empty_like(imgs[0])
    for k, img in enumerate(imgs):
        buf[...] = img
        c.SiftExtractHost(k, buf)
        buf[...] = 0
    for k in range(len(imgs)):
        kp, desc = c.GetFeatures(k)
        assert np.array_equal(kp.view(np.uint8), ref[k][0].view(np.uint8)) and np.array_equal(desc, ref[k][1]), f"frame {k}"
    c.close()


def test_sift_c2_strip_1920x1080(ctx, oracle):
    """BASELINE config C2 shape: frames of a 1920x1080 strip, batched extraction (octave 0 = 3840x2160 through the streamed
    blur, strips an exact multiple of 256 wide), features and the adjacent-pair results against the oracle pipeline"""
    import torch
    import imagemosaicing_amd as im
    from tests.synth_frames import strip
    frames, Hs = strip(3, 1920, 1080, seed=2)
    c = im.Context(0)
    dev = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
    torch.cuda.synchronize()
    for k, d in enumerate(dev):
        c.SiftExtractDev(k, d.data_ptr(), 1920, 1080, 1920 * 3)
    res = c.MatchPairs([(0, 1), (1, 2)], 2.5, 5)
    feats = []
    for k in range(3):
        kp, desc = c.GetFeatures(k)
        okp, odesc = oracle.sift(frames[k])
        assert np.array_equal(kp.view(np.uint8), okp.view(np.uint8)), f"frame {k}: keypoints differ"
        assert np.array_equal(desc.astype(np.uint8), odesc), f"frame {k}: descriptors differ"
        feats.append((okp, odesc))
    for p, (i, j) in enumerate([(0, 1), (1, 2)]):
        (k1, d1), (k2, d2) = feats[i], feats[j]
        nin, i1, i2, Ho, ns = oracle.match_pair(np.stack([k1["x"], k1["y"]], 1), d1, np.stack([k2["x"], k2["y"]], 1), d2, 1920, 1080, 2.5, 5)
        r = res[p]
        assert int(r["accepted"]) == 1 and nin == int(r["n_in"]) and ns == int(r["n_selected"])
        assert np.array_equal(r["a"][:nin], i1[:nin]) and np.array_equal(r["H"].view(np.uint32), Ho.view(np.uint32))
    c.close()
