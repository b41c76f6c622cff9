"""Multiband blend: properties of the CPU restatement (oracle/oracle_blend.c, PARITY UNPINNED: OpenCV
2.4.0's blender arithmetic is not available) and, on the GPU, byte parity of the HIP path against it."""
import ctypes as C
import numpy as np
import pytest

from tests.synth import mosaic_case, texture


def _chip(img):
    """BGR u8 image -> chip buffer with rows padded to 4 bytes, full mask"""
    h, w, _ = img.shape
    cws, mws = (w * 3 + 3) & ~3, (w + 3) & ~3
    c = np.zeros((h, cws), np.uint8); c[:, :w * 3] = img.reshape(h, -1)
    m = np.zeros((h, mws), np.uint8); m[:, :w] = 255
    return c, m


def _info(rects):
    from tests.oracle_lib import CHIPINFO
    a = np.zeros(len(rects), CHIPINFO)
    for k, (x0, y0, w, h) in enumerate(rects):
        a[k]["x0"], a[k]["y0"], a[k]["w"], a[k]["h"], a[k]["img"] = x0, y0, w, h, k
    return a


def test_reduce_expand_keep_constants(oracle):
    L = oracle.L
    src = np.full((64, 96, 3), 1234, np.int16)
    dn = np.zeros((32, 48, 3), np.int16)
    L.orc_pyr_down16(src.ctypes.data_as(C.c_void_p), 96, 64, dn.ctypes.data_as(C.c_void_p))
    assert (dn == 1234).all()
    up = np.zeros((64, 96, 3), np.int16)
    L.orc_pyr_up16(dn.ctypes.data_as(C.c_void_p), 48, 32, up.ctypes.data_as(C.c_void_p))
    assert (up == 1234).all()
    wf = np.full((64, 96), 0.625, np.float32)
    df = np.zeros((32, 48), np.float32)
    L.orc_pyr_down_f(wf.ctypes.data_as(C.c_void_p), 96, 64, df.ctypes.data_as(C.c_void_p))
    assert (df == np.float32(0.625)).all()


def test_blend_single_full_chip_is_nearly_identity(oracle):
    # every level truncates toward zero twice ((short)(lap * w), (short)(sum / (w + 1e-5))): a smooth image comes back within
    # 2 grey levels away from the border, a noisy one within ~12 (that loss of Laplacian magnitude is the i16 blender's own)
    smooth = np.tile((np.arange(400) // 2)[None, :, None], (300, 1, 3)).astype(np.uint8)
    for img, tol in ((smooth, 2), (texture(400, 300, seed=5), 12)):
        c, m = _chip(img)
        out, nb = oracle.multiband_blend(_info([(0, 0, 400, 300)]), [c], [m], 400, 300, band=5)
        assert nb == 5
        d = out[:, :1200].reshape(300, 400, 3).astype(int) - img.astype(int)
        assert np.abs(d[32:-32, 32:-32]).max() <= tol


def test_blend_two_chips_seam(oracle):
    """left half 60, right half 200, masks split at the seam: far from the seam the inputs survive, at the seam the step is
    spread over the low bands (monotone ramp)"""
    W, H = 256, 64
    a = np.full((H, W, 3), 60, np.uint8); b = np.full((H, W, 3), 200, np.uint8)
    ca, ma = _chip(a); cb, mb = _chip(b)
    ma[:, 128:] = 0; mb[:, :128] = 0
    out, _ = oracle.multiband_blend(_info([(0, 0, W, H), (0, 0, W, H)]), [ca, cb], [ma, mb], W, H, band=5)
    row = out[32, :W * 3:3].astype(int)
    assert abs(row[2] - 60) <= 6 and abs(row[-3] - 200) <= 6
    assert (np.diff(row) >= -1).all() and row[127] < row[128] + 1 and 60 < row[128] < 200


def test_blend_window_from_a_crop_equals_the_full_blend(oracle):
    """the property tests/test_gpu_full_size.py leans on at the C5 size: a window of the 5-band result is reproduced exactly by
    blending only the chips that reach it, cut to the window grown by 256 px on a 32 px grid (REDUCE / EXPAND reach: a 64 px
    margin is NOT enough, 128 is), with FindMasksByDistMap's ownership decided inside the crop only"""
    from tests import oracle_lib as ol
    rng = np.random.default_rng(1)
    n, w, h, S, AL = 8, 700, 500, 128, 32
    imgs = [texture(w, h, seed=10 + k) for k in range(n)]
    h9 = np.zeros((n, 9), np.float32)
    for k in range(n):
        a = np.deg2rad(rng.uniform(-10, 10)); sc = 1 + rng.uniform(-0.05, 0.05)
        h9[k] = np.array([[sc * np.cos(a), -sc * np.sin(a), rng.uniform(0, 1100)], [sc * np.sin(a), sc * np.cos(a), rng.uniform(0, 900)],
                          [rng.normal(0, 1e-6), rng.normal(0, 1e-6), 1]]).reshape(9)
    h9[0] = np.eye(3).reshape(9)
    r = oracle.chips_and_masks(imgs, h9, find_masks=True)
    full, nb = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=5)
    bw, bh = r["cw"], r["ch"]
    chips, lw, lh, dG = oracle.chip_layout([w] * n, [h] * n, h9)
    assert (lw, lh) == (bw, bh) and nb == 5

    def window(x0, y0, M):
        cx0, cy0 = max(0, (x0 - M) // AL * AL), max(0, (y0 - M) // AL * AL)
        cx1, cy1 = min(bw, x0 + S + M), min(bh, y0 + S + M)
        sub = [c for c in chips if c["x0"] < cx1 and c["x0"] + c["w"] > cx0 and c["y0"] < cy1 and c["y0"] + c["h"] > cy0]
        cm = [oracle.chip_warp(imgs[int(c["img"])], h9[int(c["img"])], dG, c) for c in sub]
        masks = [m for _, m in cm]
        sh = np.array(sub, ol.CHIPINFO); sh["x0"] -= cx0; sh["y0"] -= cy0
        oracle.find_masks_by_distmap(masks, sh, cx1 - cx0, cy1 - cy0)
        info = np.zeros(len(sub), ol.CHIPINFO); cc_, mm_ = [], []
        for q, (c, (chip, _), mask) in enumerate(zip(sh, cm, masks)):
            ax0, ay0 = max(0, -int(c["x0"])), max(0, -int(c["y0"]))
            ax1, ay1 = min(int(c["w"]), cx1 - cx0 - int(c["x0"])), min(int(c["h"]), cy1 - cy0 - int(c["y0"]))
            cw_, ch_ = ax1 - ax0, ay1 - ay0
            cc = np.zeros((ch_, (cw_ * 3 + 3) & ~3), np.uint8); cc[:, :cw_ * 3] = chip[ay0:ay1, 3 * ax0:3 * ax1]
            mm = np.zeros((ch_, (cw_ + 3) & ~3), np.uint8); mm[:, :cw_] = mask[ay0:ay1, ax0:ax1]
            info[q]["x0"], info[q]["y0"], info[q]["w"], info[q]["h"], info[q]["img"] = int(c["x0"]) + ax0, int(c["y0"]) + ay0, cw_, ch_, q
            cc_.append(cc); mm_.append(mm)
        ref, _ = oracle.multiband_blend(info, cc_, mm_, cx1 - cx0, cy1 - cy0, band=5)
        a = ref[y0 - cy0:y0 - cy0 + S, 3 * (x0 - cx0):3 * (x0 - cx0 + S)]
        return int((a != full[y0:y0 + S, 3 * x0:3 * (x0 + S)]).sum())

    for (x0, y0) in ((300, 250), (700, 500), (40, 30), (bw - S - 12, bh - S - 12)):
        assert window(x0, y0, 256) == 0
    assert window(700, 500, 32) > 0          # the check can fail: too small a margin changes the window


def test_the_binarys_float_reduce_order_moves_bytes_by_one_level_at_most(oracle):
    """opencv_imgproc240.dll adds the four terms of a float REDUCE value in an order that depends on the loop that produces it
    (oracle_blend.c orc_pyr_down_f: border table / 4 x unrolled slots / remainder; SSE rows / scalar tail); oracle and product keep the
    published source's single order.  Measured here on a 12-chip mosaic with every association of the binary: the weights of levels 1-3
    are the same bits (their sums are exact in float whatever the order), those of levels 4-5 differ by a few ulp, and the blended image
    differs by ONE grey level at most, in 0.3 % of its bytes (a +-1 of a level-4 / 5 Laplacian spreads over 16 / 32 pixels)."""
    rng = np.random.default_rng(7)
    n, w, h = 12, 640, 480
    imgs = [texture(w, h, seed=40 + k) for k in range(n)]
    h9 = np.zeros((n, 9), np.float32)
    for k in range(n):
        a = np.deg2rad(rng.uniform(-12, 12)); sc = 1 + rng.uniform(-0.05, 0.05)
        h9[k] = np.array([[sc * np.cos(a), -sc * np.sin(a), rng.uniform(0, 1500)], [sc * np.sin(a), sc * np.cos(a), rng.uniform(0, 1100)],
                          [rng.normal(0, 1e-6), rng.normal(0, 1e-6), 1]]).reshape(9)
    h9[0] = np.eye(3).reshape(9)
    r = oracle.chips_and_masks(imgs, h9, find_masks=True)
    import ctypes as C
    # the REDUCE alone: a 0 / 1 mask goes down five levels in both orders
    m = (rng.random((480, 640)) < 0.5).astype(np.float32)
    m[:, :200] = 1.0; m[300:, :] = 0.0
    levels = {}
    for mode in (0, 1):
        oracle.L.orc_set_float_reduce_mode(mode)
        cur, out = m, []
        for _ in range(5):
            d = np.zeros((cur.shape[0] // 2, cur.shape[1] // 2), np.float32)
            oracle.L.orc_pyr_down_f(cur.ctypes.data_as(C.c_void_p), cur.shape[1], cur.shape[0], d.ctypes.data_as(C.c_void_p))
            out.append(d); cur = d
        levels[mode] = out
    oracle.L.orc_set_float_reduce_mode(0)
    for l in range(3):
        assert np.array_equal(levels[0][l].view(np.uint32), levels[1][l].view(np.uint32)), l      # exact sums: any order gives the same bits
    ulp = [int((levels[0][l].view(np.int32).astype(np.int64) - levels[1][l].view(np.int32)).__abs__().max()) for l in (3, 4)]
    assert max(ulp) >= 1 and max(ulp) <= 4, ulp                                                   # levels 4-5: last bits
    try:
        a, nb = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=5)
        oracle.L.orc_set_float_reduce_mode(1)
        b, _ = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=5)
    finally:
        oracle.L.orc_set_float_reduce_mode(0)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert nb == 5 and d.max() == 1 and (d != 0).sum() <= 1e-2 * d.size, (int(d.max()), int((d != 0).sum()), d.size)      # measured: 25 381 of 9 258 880 bytes


@pytest.mark.gpu
def test_gpu_blend_vs_oracle(oracle):
    import imagemosaicing_amd as im
    ctx = im.Context(0)
    imgs, h9s = mosaic_case()
    r = ctx.ChipsAndMasks(imgs, h9s, find_masks=True)
    for band in (5, 2, 0):
        ref, nb = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=band)
        got, ow, oh, ows = ctx.MultiBandBlend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=band)
        assert (ow, oh) == (r["cw"], r["ch"]) and got.shape == ref.shape
        assert np.array_equal(got, ref), f"band {band}: {(got != ref).sum()} bytes differ"
    # masks that overlap / leave holes (valid-pixel masks instead of the distance-map partition)
    r2 = ctx.ChipsAndMasks(imgs, h9s, find_masks=False)
    ref, _ = oracle.multiband_blend(r2["chips"], r2["chip_imgs"], r2["masks"], r2["cw"], r2["ch"], band=5)
    got, _, _, _ = ctx.MultiBandBlend(r2["chips"], r2["chip_imgs"], r2["masks"], r2["cw"], r2["ch"], band=5)
    assert np.array_equal(got, ref)
    # one-call form: chips and masks stay in HBM between the stages, same bytes
    ref, _ = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=5)
    got, ow, oh, _ = ctx.MosaicBlended(imgs, h9s, band=5)
    assert (ow, oh) == (r["cw"], r["ch"]) and np.array_equal(got, ref)
    keep = np.array([1, 0, 1, 1], np.uint8)
    rk = ctx.ChipsAndMasks(imgs, h9s, keep=keep, find_masks=True)
    refk, _ = oracle.multiband_blend(rk["chips"], rk["chip_imgs"], rk["masks"], rk["cw"], rk["ch"], band=5)
    gotk, _, _, _ = ctx.MosaicBlended(imgs, h9s, keep=keep, band=5)
    assert np.array_equal(gotk, refk)
    ctx.close()


@pytest.mark.gpu
def test_gpu_blend_active_windows_dense_survey(oracle):
    """mi355_mosaic_blended works only inside each chip's active windows (the cell of the mosaic the chip owns + the reach of the REDUCE
    filter per level; blend.hip chip_windows) and leaves out chips that own nothing.  A dense pile of small frames -- heavy overlap, cells of
    a few dozen pixels, chips that own nothing at all, odd window origins -- must give the bytes of the full computation (the oracle works
    on whole regions), for several pyramid depths."""
    import imagemosaicing_amd as im
    rng = np.random.default_rng(20260930)
    ctx = im.Context(0)
    for trial, (n, w, h, spread) in enumerate([(24, 320, 240, 260.0), (40, 200, 152, 120.0), (9, 413, 307, 500.0)]):
        imgs, h9s = [], []
        for k in range(n):
            imgs.append(texture(w, h, seed=100 * trial + k))
            yaw = np.deg2rad(rng.uniform(-8, 8)); s = 1 + rng.uniform(-0.03, 0.03)
            tx, ty = (0.0, 0.0) if k == 0 else rng.uniform(0, spread, 2)
            H = np.array([[s * np.cos(yaw), -s * np.sin(yaw), tx], [s * np.sin(yaw), s * np.cos(yaw), ty], [0, 0, 1]], np.float32)
            h9s.append(H.reshape(9))
        h9s = np.stack(h9s)
        # one frame exactly on top of another one: the later of the two can never own a pixel (strict maximum, first wins)
        h9s[n - 1] = h9s[n // 2]; imgs[n - 1] = imgs[n // 2].copy()
        r = ctx.ChipsAndMasks(imgs, h9s, find_masks=True)
        owned = [int((m != 0).sum()) for m in r["masks"]]
        assert min(owned) == 0 and max(owned) > 0
        for band in (5, 3, 1):
            ref, _ = oracle.multiband_blend(r["chips"], r["chip_imgs"], r["masks"], r["cw"], r["ch"], band=band)
            got, ow, oh, _ = ctx.MosaicBlended(imgs, h9s, band=band)
            assert (ow, oh) == (r["cw"], r["ch"])
            assert np.array_equal(got, ref), f"trial {trial} band {band}: {(got != ref).sum()} bytes differ"
    ctx.close()


@pytest.mark.gpu
def test_gpu_blend_stripes_equal_whole_canvas():
    """mi355_mosaic_blended_rows_dev: a rank's stripe of LaplacianPyramidBlending (the default compositing path, blending = 2).  Stripes put
    side by side must be the whole canvas byte for byte, whatever the cut: eight even stripes, single rows, cuts that are odd, that fall
    inside one 2^band block, at the canvas's first and last rows -- on dense piles of small frames (cells of a few dozen pixels, chips that
    own nothing, chips that reach several stripes) and for several pyramid depths.  (The whole canvas equals the oracle's:
    test_gpu_blend_active_windows_dense_survey, same surveys.)"""
    import torch
    import imagemosaicing_amd as im
    rng = np.random.default_rng(20261001)
    ctx = im.Context(0)
    dev = torch.device("cuda", 0)
    for trial, (n, w, h, spread) in enumerate([(24, 320, 240, 260.0), (40, 200, 152, 120.0), (9, 413, 307, 500.0), (30, 256, 192, 900.0)]):
        imgs, h9s = [], []
        for k in range(n):
            imgs.append(texture(w, h, seed=300 * trial + k))
            yaw = np.deg2rad(rng.uniform(-8, 8)); s = 1 + rng.uniform(-0.03, 0.03)
            tx, ty = (0.0, 0.0) if k == 0 else rng.uniform(0, spread, 2)
            H = np.array([[s * np.cos(yaw), -s * np.sin(yaw), tx], [s * np.sin(yaw), s * np.cos(yaw), ty], [0, 0, 1]], np.float32)
            h9s.append(H.reshape(9))
        h9s = np.stack(h9s)
        h9s[n - 1] = h9s[n // 2]; imgs[n - 1] = imgs[n // 2].copy()
        d_imgs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in imgs]
        ptrs = [t.data_ptr() for t in d_imgs]
        wv, hv, wsv = [w] * n, [h] * n, [imgs[0].strides[0]] * n
        keep = None if trial != 1 else (np.arange(n) % 7 != 3).astype(np.uint8)
        for band in (5, 3, 1):
            whole, cw, ch, cws = ctx.MosaicBlendedDev(ptrs, wv, hv, wsv, h9s, keep=keep, band=band)
            whole = whole.cpu().numpy()
            cuts = [list(np.linspace(0, ch, 9).astype(int)),                                    # eight stripes
                    [0, 1, 2, 33, 64, 65, ch // 2 - 1, ch // 2, ch - 31, ch - 1, ch],            # single rows, odd cuts, cuts inside a block
                    sorted(set([0, ch] + [int(v) for v in rng.integers(1, ch, 5)]))]
            for cut in cuts:
                cut = sorted(set(int(c) for c in cut if 0 <= c <= ch))
                for a, b in zip(cut[:-1], cut[1:]):
                    got, cw2, ch2, cws2 = ctx.MosaicBlendedDev(ptrs, wv, hv, wsv, h9s, keep=keep, band=band, row0=a, rows=b - a)
                    assert (cw2, ch2, cws2) == (cw, ch, cws) and tuple(got.shape) == (b - a, cws)
                    got = got.cpu().numpy()
                    assert np.array_equal(got, whole[a:b]), f"trial {trial} band {band} rows {a}..{b - 1}: {int((got != whole[a:b]).sum())} bytes differ"
    ctx.close()
