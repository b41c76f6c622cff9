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
