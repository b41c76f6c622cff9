"""CPU, build container only: the oracle restatement against the reference's own code
(oracle/_ref/libref_oracle.so) on fresh random inputs -- wider than the committed vectors.
Skipped where oracle/_ref is absent and /root/reference cannot rebuild it."""
import numpy as np
from tests.golden_util import bits
from tests.synth import synth_pairs, texture


def test_inverse_random(oracle, ref):
    rng = np.random.default_rng(0)
    for order in (2, 3, 8, 13):
        for eps in (1e-6, 1e-12, 1e-20):
            for _ in range(30):
                a = (rng.normal(size=(order, order)) * rng.choice([1, 100, 1e4])).astype(np.float32)
                r1, o1 = oracle.inverse_matrix(a, eps)
                r2, o2 = ref.inverse_matrix(a, eps)
                assert r1 == r2
                if r1 == 1:
                    assert np.array_equal(bits(o1), bits(o2))


def test_ransac_random(oracle, ref):
    # beyond the live path's maxNum = 400 too: Ransac2D accepts any n (mosaicimage.h:1729-1761), mi355_ransac2d up to 65535
    for n, of in [(60, 0.4), (396, 0.3), (396, 0.6), (10, 0.2), (250, 0.9), (401, 0.5), (1500, 0.6), (4096, 0.7), (9001, 0.4), (30000, 0.8)]:
        for seed in (11, 12):
            p1, p2 = synth_pairs(n, of, seed=seed * 31 + n, size=(4000, 3000))
            a = oracle.ransac2d(p1, p2, 2.5, 1000, seed)
            b = ref.ransac2d(p1, p2, 2.5, 1000, seed)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            if len(a[1]) >= 4:
                assert np.array_equal(bits(a[3]), bits(b[3]))


def test_warp_random(oracle, ref):
    rng = np.random.default_rng(4)
    img = texture(200, 150, seed=8)
    for _ in range(6):
        H = np.eye(3) + rng.normal(0, 0.05, (3, 3))
        H[0, 2] = rng.uniform(-40, 40); H[1, 2] = rng.uniform(-40, 40)
        H[2, 0] = rng.normal(0, 1e-4); H[2, 1] = rng.normal(0, 1e-4); H[2, 2] = 1
        h9 = H.reshape(9).astype(np.float32)
        r1, a = oracle.image_projection_transform(img, h9)
        r2, b = ref.image_projection_transform(img, h9)
        assert r1 == r2 and a[1:] == b[1:] and np.array_equal(a[0], b[0])


def test_ransac_degenerate_inputs(oracle, ref):
    """the degenerate correspondences of tests/test_gpu_parity.py::test_ransac2d_degenerate_inputs_vs_oracle, oracle vs the
    reference's own code"""
    import numpy as np
    from tests.test_gpu_parity import _degenerate_case
    rng = np.random.default_rng(77)
    for kind in range(5):
        for m in (4, 9, 60, 250, 400):
            p1, p2 = _degenerate_case(rng, kind, m)
            seed = int(rng.integers(1, 1 << 31)); st = int(rng.choice([1000, 200]))
            a = oracle.ransac2d(p1, p2, 2.5, st, seed)
            b = ref.ransac2d(p1, p2, 2.5, st, seed)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (kind, m)
            assert np.array_equal(bits(a[3]), bits(b[3])), (kind, m)
