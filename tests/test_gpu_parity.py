"""GPU parity tests proper: the HIP path, called through the C ABI (libmi355mosaic.so via ctypes),
against (a) the golden vectors the reference's own code produced and (b) the CPU oracle on seeded inputs.
Everything here is bit-exact: float results are compared by bit pattern."""
import numpy as np
import pytest

from tests.golden_util import math_golden, warp_golden, bits
from tests.synth import synth_pairs, texture, warp_cases, mosaic_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import imagemosaicing_amd as im
    c = im.Context(0)
    yield c
    c.close()


def _rand_desc(rng, n):
    """SIFT-like integer descriptors: sparse-ish 0..255 values with clipped peaks."""
    d = rng.gamma(0.6, 25.0, size=(n, 128))
    return np.clip(d, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- warps
def test_image_projection_transform_golden(ctx):
    g = warp_golden()
    img = texture(320, 240, seed=3)
    for k, H in enumerate(warp_cases()):
        buf, dw, dh, dws = ctx.ImageProjectionTransform(img, H)
        assert [dw, dh, dws] == g[f"ipt{k}_dims"].tolist()
        assert np.array_equal(buf, g[f"ipt{k}"]), f"case {k}: {(buf != g[f'ipt{k}']).sum()} bytes differ"
    gray = np.ascontiguousarray(img[..., 1])
    buf, dw, dh, dws = ctx.ImageProjectionTransform(gray, warp_cases()[3])
    assert [dw, dh, dws] == g["ipt_gray_dims"].tolist() and np.array_equal(buf, g["ipt_gray"])


def test_mosaic_images_refined_golden(ctx):
    g = warp_golden()
    imgs, h9s = mosaic_case()
    canvas, cw, ch, cws = ctx.MosaicImagesRefined(imgs, h9s)
    assert [cw, ch, cws] == g["mosaic_dims"].tolist()
    assert np.array_equal(canvas, g["mosaic"])
    h9s[2, 8] = 0
    canvas, cw, ch, cws = ctx.MosaicImagesRefined(imgs, h9s)
    assert [cw, ch, cws] == g["mosaic_skip_dims"].tolist() and np.array_equal(canvas, g["mosaic_skip"])


def test_warp_vs_oracle_random(ctx, oracle):
    rng = np.random.default_rng(11)
    for (w, h) in [(640, 480), (333, 257)]:
        img = texture(w, h, seed=w)
        for _ in range(4):
            H = np.eye(3) + rng.normal(0, 0.06, (3, 3))
            H[0, 2] = rng.uniform(-60, 60); H[1, 2] = rng.uniform(-60, 60)
            H[2, 0] = rng.normal(0, 1e-4); H[2, 1] = rng.normal(0, 1e-4); H[2, 2] = 1
            h9 = H.reshape(9).astype(np.float32)
            rc, ref = oracle.image_projection_transform(img, h9)
            buf, dw, dh, dws = ctx.ImageProjectionTransform(img, h9)
            assert (dw, dh, dws) == ref[1:] and np.array_equal(buf, ref[0])


def test_mosaic_stripes_equal_whole(ctx):
    """canvas stripes (multi-GPU decomposition) reproduce the monolithic canvas"""
    import torch
    import imagemosaicing_amd as im
    imgs, h9s = mosaic_case()
    whole, cw, ch, cws = ctx.MosaicImagesRefined(imgs, h9s)
    d_imgs = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    w = [i.shape[1] for i in imgs]; h = [i.shape[0] for i in imgs]; ws = [i.strides[0] for i in imgs]
    lw, lh, lws, dG = im.mosaic_layout(w, h, h9s)
    assert (lw, lh, lws) == (cw, ch, cws)
    canvas = torch.full((ch, cws), 77, dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bounds = [0, ch // 3, 2 * ch // 3, ch]
    for a, b in zip(bounds[:-1], bounds[1:]):
        ctx.MosaicImagesRefinedDev([t.data_ptr() for t in d_imgs], w, h, ws, h9s, canvas.data_ptr(), cw, ch, cws, a, b - a)
    ctx.synchronize()
    ctx.set_stream(None)
    assert np.array_equal(canvas.cpu().numpy(), whole)


def test_mosaic_many_overlapping_images_vs_oracle(ctx, oracle):
    """the one-launch canvas (every tile walks its covering images in descending index and keeps the first valid sample) against the
    oracle's image-after-image overwrite: heavy overlap, projective maps, skipped images (m8 = 0), canvas widths that are not
    multiples of 4 or of the tile, images larger than the staging window's reach (strong scale), and odd stripes"""
    import torch
    import imagemosaicing_amd as im
    rng = np.random.default_rng(5)
    for trial, (n, w, h, spread, sc) in enumerate([(14, 200, 150, 160, 0.05), (9, 333, 257, 500, 0.3), (5, 640, 480, 300, 0.02), (40, 96, 64, 150, 0.1)]):
        imgs = [texture(w, h, seed=100 * trial + k) for k in range(n)]
        h9s = np.zeros((n, 9), np.float32)
        for k in range(n):
            H = np.eye(3) + rng.normal(0, sc, (3, 3))
            H[0, 2] = rng.uniform(-spread, spread); H[1, 2] = rng.uniform(-spread, spread)
            H[2, 0] = rng.normal(0, 2e-4) if k % 3 else 0.0; H[2, 1] = rng.normal(0, 2e-4) if k % 3 else 0.0; H[2, 2] = 1
            h9s[k] = H.reshape(9)
        h9s[0] = np.eye(3).reshape(9)
        if n > 6:
            h9s[3, 8] = 0; h9s[n - 1, 8] = 0          # invalid images are skipped (MosaicWithoutPos.cpp:2256)
        rc, (ref, rw, rh, rws) = oracle.mosaic_images_refined(imgs, h9s)
        assert rc == 0
        got, cw, ch, cws = ctx.MosaicImagesRefined(imgs, h9s)
        assert (cw, ch, cws) == (rw, rh, rws)
        assert np.array_equal(got, ref), f"trial {trial}: {int((got != ref).sum())} bytes differ"
        d_imgs = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        ws_ = [i.strides[0] for i in imgs]
        canvas = torch.full((ch, cws), 201, dtype=torch.uint8, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        cuts = sorted(set([0, ch] + [int(x) for x in rng.integers(1, ch, 4)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.MosaicImagesRefinedDev([t.data_ptr() for t in d_imgs], [w] * n, [h] * n, ws_, h9s, canvas.data_ptr(), cw, ch, cws, a, b - a)
        ctx.synchronize()
        ctx.set_stream(None)
        assert np.array_equal(canvas.cpu().numpy(), ref), f"trial {trial}: stripes differ"


def test_chips_and_masks_vs_oracle(ctx, oracle):
    imgs, h9s = mosaic_case()
    ref = oracle.chips_and_masks(imgs, h9s, find_masks=True)
    got = ctx.ChipsAndMasks(imgs, h9s, find_masks=True)
    assert (got["cw"], got["ch"]) == (ref["cw"], ref["ch"])
    assert len(got["chips"]) == len(ref["chips"])
    for k in range(len(ref["chips"])):
        for f in ("x0", "y0", "w", "h", "img"):
            assert int(got["chips"][k][f]) == int(ref["chips"][k][f])
        assert np.array_equal(bits(got["chips"][k]["quad"]), bits(ref["chips"][k]["quad"]))
        assert np.array_equal(got["chip_imgs"][k], ref["chip_imgs"][k])
        assert np.array_equal(got["masks"][k], ref["masks"][k])
    got_v = ctx.ChipsAndMasks(imgs, h9s, find_masks=False)
    for k in range(len(ref["chips"])):
        assert np.array_equal(got_v["masks"][k], ref["valid"][k])


def test_chips_and_masks_golden_from_reference(ctx, lib):
    """chips, validity and distance-map ownership masks against the vectors the reference's OWN code produced
    (LaplacianPyramidBlending warp stage MosaicImage.cpp:2216-2460 + FindMasksByDistMap :1761-1881 through oracle/_ref)"""
    from tests.golden_util import blend_golden
    from tests.test_overlap import blend_cases, overlap_layouts
    g = blend_golden()
    for tag, imgs, h9s in blend_cases():
        got = ctx.ChipsAndMasks(imgs, h9s, find_masks=True)
        assert [got["cw"], got["ch"], len(got["chips"])] == g[f"{tag}_dims"].tolist()
        for k in range(len(got["chips"])):
            assert [int(got["chips"][k][f]) for f in ("x0", "y0", "w", "h", "img")] == g[f"{tag}_geom{k}"].tolist()
            assert np.array_equal(bits(got["chips"][k]["quad"]), g[f"{tag}_quad{k}"])
            assert np.array_equal(got["chip_imgs"][k], g[f"{tag}_chip{k}"]), (tag, k)
            assert np.array_equal(got["masks"][k], g[f"{tag}_mask{k}"]), (tag, k)
    # ResampleByOverlap's decision feeds the same call as keep[]
    for k, (w, h, h9) in enumerate(overlap_layouts(7)):
        assert np.array_equal(lib.resample_by_overlap(w, h, h9, 0.7), g[f"keep{k}"])


# ---------------------------------------------------------------------------------------------- RANSAC
def test_ransac2d_golden(ctx):
    g = math_golden()
    for p1, p2, n, seed, ok, nin, ids, H in zip(g["r_p1"], g["r_p2"], g["r_n"], g["r_seed"], g["r_ok"], g["r_nin"], g["r_ids"], g["r_H"]):
        ok2, i1, i2, H2 = ctx.Ransac2D(p1[:n].copy(), p2[:n].copy(), 2.5, 1000, int(seed))
        assert ok2 == ok and len(i1) == nin, (n, seed, ok2, ok, len(i1), nin)
        assert np.array_equal(i1["id"], ids[:nin])
        if nin >= 4:
            assert np.array_equal(bits(H2), bits(H)), (n, seed, H2, H)


def test_ransac2d_one_workgroup_per_pair_and_split_forms_agree(ctx, oracle):
    """Few pairs run as several workgroups per pair and three launches (ransac.hip "few pairs": classify / evaluate / finish with the wide
    closing refinement), many pairs as one workgroup per pair; option ransac_split forces either.  The golden cases from the reference's own
    code, random cases against the oracle and whole pair records (n_in, inlier lists, H bits, the padding word) must not depend on the form."""
    g = math_golden()
    try:
        for S in (0, 1, 3, 8):
            ctx.set_option("ransac_split", S)
            for p1, p2, n, seed, ok, nin, ids, H in zip(g["r_p1"], g["r_p2"], g["r_n"], g["r_seed"], g["r_ok"], g["r_nin"], g["r_ids"], g["r_H"]):
                ok2, i1, i2, H2 = ctx.Ransac2D(p1[:n].copy(), p2[:n].copy(), 2.5, 1000, int(seed))
                assert ok2 == ok and len(i1) == nin and np.array_equal(i1["id"], ids[:nin]), (S, n, seed)
                if nin >= 4:
                    assert np.array_equal(bits(H2), bits(H)), (S, n, seed, H2, H)
            for n, of, st in [(396, 0.35, 1000), (396, 0.8, 1000), (57, 0.5, 1000), (31, 0.2, 1000), (9, 0.0, 1000), (4, 0.0, 1000), (396, 0.5, 1), (396, 0.5, 77), (200, 0.6, 4999), (13, 0.5, 5000),
                              (396, 0.5, 4999), (400, 0.3, 5000)]:      # the last two: draw list + body past 80 KB of LDS -> the list lives in HBM (ransac_listg_kernel)
                p1, p2 = synth_pairs(n, of, seed=1000 + 31 * n + st, size=(4000, 3000))
                a = oracle.ransac2d(p1, p2, 2.5, st, 5)
                b = ctx.Ransac2D(p1, p2, 2.5, st, 5)
                assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (S, n, of, st)
                if len(a[1]) >= 4:
                    assert np.array_equal(bits(a[3]), bits(b[3])), (S, n, of, st)
        # whole records of a batch of pairs (accepted, rejected, degenerate), byte for byte between the forms
        rng = np.random.default_rng(4)
        for k in range(6):
            kp1, d1, kp2, d2 = _synthetic_feature_pair(rng, n=2000 if k % 2 == 0 else 500 + 100 * k)
            ctx.SetFeatures(2 * k, kp1, d1.astype(np.float32), 4000, 3000)
            ctx.SetFeatures(2 * k + 1, kp2, d2.astype(np.float32), 4000, 3000)
        pairs = [(2 * k, 2 * k + 1) for k in range(6)] + [(0, 3), (2, 7), (1, 2)]
        recs = []
        for S in (0, 2, 8):
            ctx.set_option("ransac_split", S)
            recs.append(ctx.MatchPairs(pairs, 2.5, 99).tobytes())
        assert recs[0] == recs[1] == recs[2]
    finally:
        ctx.set_option("ransac_split", -1)
        ctx.DropFeatures(-1)


def test_ransac2d_vs_oracle_random(ctx, oracle):
    for n, of in [(396, 0.35), (396, 0.8), (123, 0.5), (9, 0.0), (4, 0.0)]:
        for seed in (21, 22):
            p1, p2 = synth_pairs(n, of, seed=seed * 13 + n, size=(4000, 3000))
            a = oracle.ransac2d(p1, p2, 2.5, 1000, seed)
            b = ctx.Ransac2D(p1, p2, 2.5, 1000, seed)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            if len(a[1]) >= 4:
                assert np.array_equal(bits(a[3]), bits(b[3]))


def test_ransac2d_more_than_400_correspondences(ctx, oracle):
    """Ransac2D accepts any n (mosaicimage.h:1729-1761); the live path stops at 396, mi355_ransac2d goes on to 4096 through a second
    kernel (work arrays of the closing Gauss-Newton in HBM) and to 65 535 -- what a 16-bit draw table addresses -- through a third
    (the points in HBM as well).  Same bits as the oracle, which tests/test_oracle_vs_ref.py shows equal to the reference's own code
    at these sizes; 65 536 is refused."""
    import imagemosaicing_amd as im
    for n, of in [(401, 0.5), (777, 0.2), (1500, 0.6), (4096, 0.7), (4096, 0.97), (4097, 0.5), (9001, 0.4), (30000, 0.8), (65535, 0.6)]:
        for seed in (31, 32):
            p1, p2 = synth_pairs(n, of, seed=seed * 7 + n, size=(4000, 3000))
            a = oracle.ransac2d(p1, p2, 2.5, 1000, seed)
            b = ctx.Ransac2D(p1, p2, 2.5, 1000, seed)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (n, of, seed, a[0], b[0], len(a[1]), len(b[1]))
            if len(a[1]) >= 4:
                assert np.array_equal(bits(a[3]), bits(b[3])), (n, of, seed)
    p1, p2 = synth_pairs(65536, 0.5, seed=5, size=(4000, 3000))
    with pytest.raises(im.Mi355Error):
        ctx.Ransac2D(p1, p2, 2.5, 1000, 1)


def test_ransac2d_edge_cases(ctx):
    from tests.oracle_lib import sfpoints
    p = sfpoints(np.zeros((3, 2)))
    ok, i1, i2, H = ctx.Ransac2D(p, p, 2.5, 1000, 1)
    assert ok == 0 and len(i1) == 0
    ok, i1, i2, H = ctx.Ransac2D(p[:0], p[:0], 2.5, 1000, 1)
    assert ok == 0


# ---------------------------------------------------------------------------------------------- selection
def test_select_match_pairs_golden(ctx):
    g = math_golden()
    for kp1, kp2, m, (w, h, K), nm, o1, o2, no in zip(g["s_kp1"], g["s_kp2"], g["s_m"], g["s_wh"], g["s_nm"], g["s_o1"], g["s_o2"], g["s_no"]):
        a1, a2 = ctx.SelectMatchPairs(m[:K], kp1[:K], kp2[:K], int(nm), int(w), int(h))
        assert len(a1) == no
        assert np.array_equal(a1, o1[:no]) and np.array_equal(a2, o2[:no])


def test_device_rand_stream_equals_glibc(ctx, oracle):
    """the RANSAC draw tables are built from a rand() stream generated on the device by jump-ahead of glibc's lagged sum (ransac.hip):
    all 400 768 values of three seeds against the oracle's restatement of glibc (itself checked against libc in test_oracle_golden)"""
    import ctypes as C
    for seed in (1, 12345, 0xFFFFFFFF):
        n = 404 * 992
        out = np.zeros(n, np.int32)
        assert ctx.L.mi355_debug_rand_stream(ctx._h, C.c_uint32(seed), out.ctypes.data_as(C.c_void_p), n) == 0
        st = (C.c_int32 * 40)()
        oracle.L.orc_srand(st, C.c_uint(seed))
        rnd = oracle.L.orc_rand
        want = np.fromiter((rnd(st) for _ in range(n)), np.int32, n)
        assert np.array_equal(out, want), int((out != want).argmax())


def test_pivoting_inverse_by_the_wave_equals_the_oracle(ctx, oracle):
    """hmath.h inverse8_wave: InverseMatrix of order 8 (matrix.h:147-296) for the draws whose J^T J needs a pivot below the diagonal, by the
    lanes of the draw's wave together (the register routine holds a lane for ~50 us per call, 15 calls per such draw).  Matrices that need row
    swaps, that have no pivot in some column (the output stays what it was), pivots at either side of eps, whole and partly filled waves (the
    last group of a list: 40 active lanes take the wave routine, 7 the register routine): the same bits as the oracle's InverseMatrix and as
    the register routine."""
    import ctypes as C
    rng = np.random.default_rng(2026)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    init = rng.normal(0, 1, 64).astype(np.float32)
    for n in (256, 64 * 3 + 40, 64 * 2 + 7, 33, 1):
        for eps in (1e-6, 1e-20):
            mats = rng.normal(0, 1, (n, 8, 8)).astype(np.float32)
            kind = rng.integers(0, 6, n)
            for i in range(n):
                m = mats[i]
                if kind[i] == 1: m[0, 0] = 0.0                                   # pivot below the diagonal in column 0
                elif kind[i] == 2: m[:, 3] = 0.0                                 # no pivot in column 3: the routine gives up
                elif kind[i] == 3: m[rng.integers(0, 8), :] = m[rng.integers(0, 8), :]      # (maybe) two equal rows
                elif kind[i] == 4: m *= np.float32(eps) * np.float32(rng.choice([0.5, 2.0]))   # entries around eps
                elif kind[i] == 5: m[:] = np.float32(np.diag(rng.normal(0, 1, 8)))[rng.permutation(8)]   # a permuted diagonal: the closing row pass
            mats = np.ascontiguousarray(mats.reshape(n, 64))
            a = np.zeros((n, 64), np.float32); b = np.zeros((n, 64), np.float32)
            assert ctx.L.mi355_debug_inverse8(ctx._h, p(mats), n, C.c_float(eps), p(init), p(a), p(b)) == 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n, eps, int((a.view(np.uint32) != b.view(np.uint32)).any(1).sum()))
            for i in range(n):
                rc, want = oracle.inverse_matrix(mats[i].reshape(8, 8), eps)
                ref = want.reshape(64) if rc == 1 else init
                assert np.array_equal(b[i].view(np.uint32), np.ascontiguousarray(ref, np.float32).view(np.uint32)), (n, eps, i, int(kind[i]), rc)


def test_guarded_division_equals_true_division(ctx):
    """hmath.h rcp_nr / div_nr (the shared-reciprocal division of the RANSAC polish): wherever the guard accepts a quotient it has the
    bits of the correctly rounded a / b -- 8 M random operand pairs over the whole exponent range, quotients next to rounding
    boundaries, and the special values; and the guard does accept the magnitudes the polish works with."""
    import ctypes as C
    rng = np.random.default_rng(77)

    def run(a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        n = len(a)
        qf = np.zeros(n, np.float32); qt = np.zeros(n, np.float32); ok = np.zeros(n, np.int32)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        assert ctx.L.mi355_debug_div(ctx._h, p(a), p(b), n, p(qf), p(qt), p(ok)) == 0
        return qf.view(np.uint32), qt.view(np.uint32), ok.astype(bool)

    n = 1 << 22
    # (1) every bit pattern is fair game: random sign / exponent / mantissa
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    qf, qt, ok = run(a, b)
    assert np.array_equal(qf[ok], qt[ok]), int((qf[ok] != qt[ok]).sum())
    # (2) the polish's magnitudes (coordinates, their products, pivots of J^T J): all accepted, all equal
    mag = lambda lo, hi: np.exp2(rng.uniform(lo, hi, n)).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)
    a, b = mag(-20, 52), mag(-18, 52)
    qf, qt, ok = run(a, b)
    inwin = (np.abs(qt.view(np.float32)) >= 2.0 ** -70) & (np.abs(qt.view(np.float32)) <= 2.0 ** 38)
    assert ok[inwin].all() and inwin.mean() > 0.7
    assert np.array_equal(qf[ok], qt[ok])
    # (3) quotients that sit next to a rounding boundary: a = round(q * b) for q with few mantissa bits, +- 1 ulp
    q = (rng.integers(1 << 23, 1 << 24, n).astype(np.float32) * np.float32(2.0 ** -23))
    b = mag(-10, 10)
    a = (q * b).astype(np.float32)
    a = np.nextafter(a, np.where(rng.random(n) < 0.5, np.float32(np.inf), np.float32(-np.inf))).astype(np.float32)
    qf, qt, ok = run(a, b)
    assert ok.mean() > 0.99 and np.array_equal(qf[ok], qt[ok])
    # (4) special values are never accepted with other bits
    sp = np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, 1.0, -1.0, 2.0 ** -100, 2.0 ** 100, 2.0 ** -31, 2.0 ** 63])
    a, b = np.repeat(sp, len(sp)), np.tile(sp, len(sp))
    qf, qt, ok = run(a, b)
    assert np.array_equal(qf[ok], qt[ok])
    assert not ok[(a == 0) | ~np.isfinite(a) | (b == 0) | ~np.isfinite(b)].any()


# ---------------------------------------------------------------------------------------------- matching
def test_bf_match_exact(ctx, oracle):
    """int8 MFMA distances (int32 accumulation) are exact integers: indices, 1-NN and 2-NN squared distances equal the CPU
    integer brute force bit for bit, including ties (duplicated descriptors) and ragged sizes."""
    rng = np.random.default_rng(5)
    from imagemosaicing_amd import KEYPOINT
    for (n1, n2) in [(2000, 2000), (1999, 1531), (130, 70), (1, 5), (64, 64), (2048, 2048), (513, 33), (7, 1), (70, 3), (33, 4)]:
        d1, d2 = _rand_desc(rng, n1), _rand_desc(rng, n2)
        if n2 > 40:
            d2[37] = d2[3]              # exact tie -> lowest train index must win
            d1[0] = d2[3]
        if n2 > 1200 and n1 > 8:
            d2[1000] = d2[3]            # the same row again, 31 tiles later: still the lowest index
            d2[5] = d2[4]; d1[1] = d2[4]                    # a tie inside one group of four rows
            d2[10] = 255; d2[11] = 0; d1[2] = 255; d1[3] = 0; d1[4] = 0; d1[4, ::2] = 255     # the ends of the int8 range, the largest norms and products
            d2[1999 if n2 > 1999 else n2 - 1] = d2[40]; d1[5] = d2[40]                        # a tie between the two half-waves' rows
        kp1 = np.zeros(n1, KEYPOINT); kp2 = np.zeros(n2, KEYPOINT)
        kp1["x"] = rng.uniform(5, 995, n1); kp1["y"] = rng.uniform(5, 745, n1)
        kp2["x"] = rng.uniform(5, 995, n2); kp2["y"] = rng.uniform(5, 745, n2)
        ctx.SetFeatures(100, kp1, d1.astype(np.float32), 1000, 750)
        ctx.SetFeatures(101, kp2, d2.astype(np.float32), 1000, 750)
        idx, b1, b2 = oracle.bf_match(d1, d2)
        m, g1, g2 = ctx.BFMatch(100, 101, sorted_=False)
        assert len(m) == n1
        assert np.array_equal(m["trainIdx"], idx) and np.array_equal(g1, b1)
        if n2 > 1:
            assert np.array_equal(g2, b2)
        else:
            assert np.all(g2 == 0x7fffffff)        # one train row: no second neighbour (lanes whose rows hold no train row report nothing)
        ms, s1, _ = ctx.BFMatch(100, 101, sorted_=True)
        want = oracle.sort_matches(idx, b1)
        assert np.array_equal(np.stack([ms["queryIdx"], ms["trainIdx"]], 1), want)
        fk, fd = ctx.GetFeatures(100)
        assert np.array_equal(fd.astype(np.uint8), d1) and np.array_equal(fk["x"], kp1["x"])
    ctx.DropFeatures(100); ctx.DropFeatures(101)


def test_bf_match_and_selection_beyond_2048_keypoints(ctx, oracle):
    """the large-pair path (keep-all frames, match.hip): query and train sets in chunks of 2048 on the matrix cores, the chunks' nearest merged
    (ties -> the lowest train index, also ACROSS chunks), the M keys sorted in HBM, the same grid walk.  1-NN / 2-NN, the sorted list and the
    stand-alone SelectMatchPairs against the CPU oracle for ragged sizes on either side of the chunk boundaries, up to the 32 768 the record holds."""
    rng = np.random.default_rng(77)
    from imagemosaicing_amd import KEYPOINT
    for (n1, n2) in [(2049, 2048), (2048, 2049), (4097, 300), (300, 4100), (5000, 6200), (9000, 3000), (32768, 2500)]:
        d1, d2 = _rand_desc(rng, n1), _rand_desc(rng, n2)
        if n2 > 2100:
            d2[2050] = d2[3]; d1[0] = d2[3]                 # an exact tie between two train CHUNKS: the lower index wins
            d2[n2 - 1] = d2[2047]; d1[1] = d2[2047]         # ... the last row of chunk 0 against the very last row
        if n1 > 2100:
            d1[2100] = d1[5]                                # equal distances for two queries of different query chunks: order by queryIdx
        kp1 = np.zeros(n1, KEYPOINT); kp2 = np.zeros(n2, KEYPOINT)
        kp1["x"] = rng.uniform(5, 995, n1); kp1["y"] = rng.uniform(5, 745, n1)
        kp2["x"] = rng.uniform(5, 995, n2); kp2["y"] = rng.uniform(5, 745, n2)
        ctx.SetFeatures(100, kp1, d1.astype(np.float32), 1000, 750)
        ctx.SetFeatures(101, kp2, d2.astype(np.float32), 1000, 750)
        idx, b1, b2 = oracle.bf_match(d1, d2)
        m, g1, g2 = ctx.BFMatch(100, 101, sorted_=False, max_matches=n1)
        assert len(m) == n1
        assert np.array_equal(m["trainIdx"], idx) and np.array_equal(g1, b1) and np.array_equal(g2, b2), (n1, n2)
        ms, s1, _ = ctx.BFMatch(100, 101, sorted_=True, max_matches=n1)
        want = oracle.sort_matches(idx, b1)
        assert np.array_equal(np.stack([ms["queryIdx"], ms["trainIdx"]], 1), want), (n1, n2)
        # stand-alone SelectMatchPairs on the whole sorted list (more than select_kernel's 2048 LDS keys when n1 > 2048)
        nMatch = int(min(400.0, 0.3 * n1))
        a1, a2 = ctx.SelectMatchPairs(want, np.stack([kp1["x"], kp1["y"]], 1), np.stack([kp2["x"], kp2["y"]], 1), nMatch, 1000, 750)
        o1, o2 = oracle.select(want, np.stack([kp1["x"], kp1["y"]], 1), np.stack([kp2["x"], kp2["y"]], 1), nMatch, 1000, 750)
        assert len(a1) == len(o1) and np.array_equal(a1, o1) and np.array_equal(a2, o2), (n1, n2)
        if n1 <= 9000:
            # the whole j-loop body on such a pair: the record equals oracle.match_pair's
            r = ctx.MatchPairs([(100, 101)], 2.5, 3)[0]
            nin, i1, i2, Ho, nsel = oracle.match_pair(np.stack([kp1["x"], kp1["y"]], 1), d1, np.stack([kp2["x"], kp2["y"]], 1), d2, 1000, 750, 2.5, 3)
            assert int(r["n_selected"]) == nsel and int(r["accepted"]) == int(nin > 30), (n1, n2)
    # a batch that mixes both kinds of pairs: every record equals the one the pair gives alone
    kpa = np.zeros(1500, KEYPOINT); kpa["x"] = rng.uniform(5, 995, 1500); kpa["y"] = rng.uniform(5, 745, 1500)
    ctx.SetFeatures(102, kpa, _rand_desc(rng, 1500).astype(np.float32), 1000, 750)
    mixed = [(102, 102), (100, 101), (102, 101), (101, 102), (102, 102), (100, 100)]
    together = ctx.MatchPairs(mixed, 2.5, 5)
    for p, r in zip(mixed, together):
        alone = ctx.MatchPairs([p], 2.5, 5)[0]
        assert np.array_equal(np.frombuffer(r.tobytes(), np.uint8), np.frombuffer(alone.tobytes(), np.uint8)), p
    ctx.DropFeatures(100); ctx.DropFeatures(101); ctx.DropFeatures(102)


def _synthetic_feature_pair(rng, n=2000, w=4000, h=3000, overlap=0.7):
    """two keypoint sets related by a homography, matching descriptors for the shared part + noise"""
    from imagemosaicing_amd import KEYPOINT
    H = np.array([1.01, 0.02, 300, -0.015, 0.99, -200, 2e-6, -3e-6, 1.0])
    kp2 = np.zeros(n, KEYPOINT); kp1 = np.zeros(n, KEYPOINT)
    kp2["x"] = rng.uniform(5, w - 5, n); kp2["y"] = rng.uniform(5, h - 5, n)
    d = H[6] * kp2["x"] + H[7] * kp2["y"] + 1
    x1 = (H[0] * kp2["x"] + H[1] * kp2["y"] + H[2]) / d + rng.normal(0, 0.4, n)
    y1 = (H[3] * kp2["x"] + H[4] * kp2["y"] + H[5]) / d + rng.normal(0, 0.4, n)
    d2 = _rand_desc(rng, n)
    d1 = np.clip(d2.astype(np.int32) + rng.integers(-6, 7, d2.shape), 0, 255).astype(np.uint8)
    ns = int(overlap * n)
    inside = (x1 > 5) & (x1 < w - 5) & (y1 > 5) & (y1 < h - 5)
    shared = np.where(inside)[0][:ns]
    kp1["x"] = rng.uniform(5, w - 5, n); kp1["y"] = rng.uniform(5, h - 5, n)
    d1r = _rand_desc(rng, n)
    perm = rng.permutation(n)
    tgt = perm[:len(shared)]
    kp1["x"][tgt] = x1[shared]; kp1["y"][tgt] = y1[shared]
    d1r[tgt] = d1[shared]
    return kp1, d1r, kp2, d2


def test_match_pairs_vs_oracle(ctx, oracle):
    """whole j-loop body (match -> sort -> grid select -> Ransac2D -> accept) on device-resident features"""
    rng = np.random.default_rng(77)
    feats = []
    for k in range(3):
        kp1, d1, kp2, d2 = _synthetic_feature_pair(rng, n=2000 if k < 2 else 777)
        feats.append((kp1, d1, kp2, d2))
        ctx.SetFeatures(2 * k, kp1, d1.astype(np.float32), 4000, 3000)
        ctx.SetFeatures(2 * k + 1, kp2, d2.astype(np.float32), 4000, 3000)
    pairs = [(0, 1), (2, 3), (4, 5), (0, 3)]       # the last one is an unrelated pair -> rejected
    seed = 4242
    res = ctx.MatchPairs(pairs, 2.5, seed)
    for r, (i, j) in zip(res, pairs):
        kp1, d1 = feats[i // 2][0], feats[i // 2][1]
        kp2, d2 = feats[j // 2][2], feats[j // 2][3]
        xy1 = np.stack([kp1["x"], kp1["y"]], 1); xy2 = np.stack([kp2["x"], kp2["y"]], 1)
        nin, i1, i2, H, ns = oracle.match_pair(xy1, d1, xy2, d2, 4000, 3000, 2.5, seed)
        assert (int(r["i"]), int(r["j"])) == (i, j)
        assert int(r["n_selected"]) == ns
        n_in = int(r["n_in"])
        assert (n_in if n_in > 30 else 0) == nin and int(r["accepted"]) == (1 if nin > 0 else 0)
        if nin > 0:
            assert np.array_equal(r["a"][:n_in], i1[:n_in]) and np.array_equal(r["b"][:n_in], i2[:n_in])
            assert np.array_equal(bits(r["H"]), bits(H))
    assert int(res[0]["accepted"]) == 1 and int(res[3]["accepted"]) == 0
    ctx.DropFeatures(-1)


def test_match_pairs_ratio_option_vs_oracle(oracle):
    """north_star option: Lowe ratio test before the grid walk (mi355_params.ratio)"""
    import imagemosaicing_amd as im
    p = im.default_params()
    p.ratio = 0.8
    c = im.Context(0, p)
    rng = np.random.default_rng(3)
    kp1, d1, kp2, d2 = _synthetic_feature_pair(rng, n=1500)
    c.SetFeatures(0, kp1, d1.astype(np.float32), 4000, 3000)
    c.SetFeatures(1, kp2, d2.astype(np.float32), 4000, 3000)
    r = c.MatchPairs([(0, 1)], 2.5, 9)[0]
    xy1 = np.stack([kp1["x"], kp1["y"]], 1); xy2 = np.stack([kp2["x"], kp2["y"]], 1)
    nin, i1, i2, H, ns = oracle.match_pair_ratio(xy1, d1, xy2, d2, 4000, 3000, 2.5, 9, 0.8)
    assert int(r["n_selected"]) == ns and ns > 0
    n_in = int(r["n_in"])
    assert (n_in if n_in > 30 else 0) == nin
    assert np.array_equal(r["a"][:n_in], i1[:n_in]) and np.array_equal(bits(r["H"]), bits(H))
    # the ratio test removes the unmatched third of the keypoints: fewer selected than without it
    nin0, _, _, _, ns0 = oracle.match_pair(xy1, d1, xy2, d2, 4000, 3000, 2.5, 9)
    assert ns <= ns0
    c.close()


def test_match_pairs_degenerate_inputs(ctx):
    """empty / tiny feature sets: no crash, pair rejected (the reference appends nothing, MosaicWithoutPos.cpp:5222-5227)"""
    from imagemosaicing_amd import KEYPOINT
    rng = np.random.default_rng(8)
    kp = np.zeros(3, KEYPOINT); kp["x"] = [10, 20, 30]; kp["y"] = [5, 6, 7]
    d = _rand_desc(rng, 3).astype(np.float32)
    ctx.SetFeatures(50, kp, d, 640, 480)
    ctx.SetFeatures(51, kp[:0], d[:0], 640, 480)
    ctx.SetFeatures(52, kp, d, 640, 480)
    res = ctx.MatchPairs([(50, 51), (51, 50), (50, 52)], 2.5, 1)
    assert (res["accepted"] == 0).all() and (res["n_in"] <= 3).all()
    with pytest.raises(Exception):
        ctx.MatchPairs([(50, 99)], 2.5, 1)          # unknown image id -> MI355_ERR_ARG
    ctx.DropFeatures(-1)


def test_chips_keep_and_invalid_flags(ctx, oracle):
    imgs, h9s = mosaic_case()
    keep = np.array([1, 0, 1, 1], np.uint8)
    h9s[2, 8] = 0          # invalid image convention
    ref = oracle.chips_and_masks(imgs, h9s, keep=keep, find_masks=True)
    got = ctx.ChipsAndMasks(imgs, h9s, keep=keep, find_masks=True)
    assert [int(c["img"]) for c in got["chips"]] == [int(c["img"]) for c in ref["chips"]] == [0, 3]
    for k in range(2):
        assert np.array_equal(got["chip_imgs"][k], ref["chip_imgs"][k]) and np.array_equal(got["masks"][k], ref["masks"][k])


def test_ransac2d_polish_that_diverges_keeps_its_slot(ctx, oracle):
    """A hypothesis whose 4-point residual is below 5 px is polished and then ALWAYS consumes a slot of the sample_times
    budget, even when the polish drives its residual past 5 (mosaicimage.h:1864-1876 tests the residual of the solve only).
    Found by scratch/soak_ransac.py: 359 garbage correspondences, where the winner changes between sample_times 200 and 201.
    Expected values: the reference's own code (tests/golden/ransac_polish_diverges.npz, made with oracle/_ref)."""
    import os
    from tests.golden_util import GOLD
    g = np.load(os.path.join(GOLD, "ransac_polish_diverges.npz"))
    for st in (200, 201):
        ok, i1, i2, H = ctx.Ransac2D(g["p1"], g["p2"], float(g["dist"]), st, int(g["seed"]))
        assert ok == int(g[f"ok{st}"]) and np.array_equal(i1, g[f"i1_{st}"]) and np.array_equal(i2, g[f"i2_{st}"]), st
        assert np.array_equal(bits(H), bits(g[f"H{st}"])), st
        a = oracle.ransac2d(g["p1"], g["p2"], float(g["dist"]), st, int(g["seed"]))
        assert a[0] == ok and np.array_equal(a[1], i1) and np.array_equal(bits(a[3]), bits(H))
    assert not np.array_equal(g["H200"], g["H201"])


def _degenerate_case(rng, kind, m):
    from tests.oracle_lib import SFPOINT
    w, h = 4000.0, 3000.0
    p1 = np.zeros(m, SFPOINT); p2 = np.zeros(m, SFPOINT)
    p2["x"] = rng.uniform(0, w, m); p2["y"] = rng.uniform(0, h, m)
    A = np.array([[1 + rng.normal(0, .02), rng.normal(0, .02), rng.uniform(-800, 800)], [rng.normal(0, .02), 1 + rng.normal(0, .02), rng.uniform(-600, 600)], [0, 0, 1]])
    q = A @ np.stack([p2["x"], p2["y"], np.ones(m)])
    p1["x"] = q[0] + rng.normal(0, 0.4, m); p1["y"] = q[1] + rng.normal(0, 0.4, m)
    out = rng.random(m) < 0.5
    p1["x"][out] = rng.uniform(0, w, out.sum()); p1["y"][out] = rng.uniform(0, h, out.sum())
    if kind == 0:      # one source point matched by several targets: singular 4-point systems, failed inversions
        k = rng.integers(0, m, m // 2); p2["x"][: m // 2] = p2["x"][k]; p2["y"][: m // 2] = p2["y"][k]
    elif kind == 1:    # duplicated pairs
        k = rng.integers(0, m, m // 3); p2[: m // 3] = p2[k]; p1[: m // 3] = p1[k]
    elif kind == 2:    # integer lattice: collinear quadruples, exact zeros in the eliminations (pivot search below the diagonal)
        p2["x"] = np.round(p2["x"] / 50) * 50; p2["y"] = np.round(p2["y"] / 50) * 50; p1["x"] = np.round(p1["x"]); p1["y"] = np.round(p1["y"])
    elif kind == 3:    # every source point on one line
        p2["y"] = 0.5 * p2["x"] + 10
    elif kind == 4:    # coordinates whose products overflow binary32: non-finite design matrices (the generic routines)
        p1["x"] *= 1e18; p1["y"] *= 1e18; p2["x"] *= 1e18; p2["y"] *= 1e18
    return p1, p2


def test_ransac2d_degenerate_inputs_vs_oracle(ctx, oracle):
    """duplicated, collinear, lattice and overflowing correspondences drive the 4-point solve through its three tiers (sparse
    register elimination, register elimination with the reference's pivot search, generic routines); every outcome -- also
    'no homography' -- must be the oracle's (= the reference's: tests/test_oracle_vs_ref.py runs the same inputs on CPU)"""
    rng = np.random.default_rng(77)
    for kind in range(5):
        for m in (4, 9, 60, 250, 400):
            p1, p2 = _degenerate_case(rng, kind, m)
            seed = int(rng.integers(1, 1 << 31)); st = int(rng.choice([1000, 200]))
            a = oracle.ransac2d(p1, p2, 2.5, st, seed)
            b = ctx.Ransac2D(p1, p2, 2.5, st, seed)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (kind, m, a[0], b[0], len(a[1]), len(b[1]))
            assert np.array_equal(bits(a[3]), bits(b[3])), (kind, m)
