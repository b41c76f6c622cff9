"""The alignment's input as second moments per accepted pair (mi355_pair_moments, round 5): 184 bytes instead of the 9664-byte record.  The
sums are the ones mi355_global_affine_align_results forms from the inlier lists (BundleAdjustmentSparse's normal equations,
MosaicWithoutPos.cpp:6971-7202), so transforms and labels must come out the same BITS from the moments -- formed on the host here, on the device
in the gpu test -- as from the records."""
import numpy as np
import pytest


def synthetic_records(im, seed, N, extra=()):
    rng = np.random.default_rng(seed)
    pairs = [(i, j) for i in range(N) for j in range(i + 1, min(N, i + 182)) if (j == i + 1 or rng.random() < 0.03)] + list(extra)
    r = np.zeros(len(pairs), im.PAIR_RESULT)
    pos = np.cumsum(rng.uniform(300, 900, (N, 2)), axis=0)
    for k, (i, j) in enumerate(pairs):
        n = int(rng.choice([31, 400, int(rng.integers(32, 400))]))
        xy = rng.uniform(0, 4000, (n, 2)).astype(np.float32)
        r["i"][k] = i; r["j"][k] = j; r["n_in"][k] = n; r["accepted"][k] = 1; r["ok"][k] = 1; r["n_selected"][k] = n
        r["a"]["x"][k, :n] = xy[:, 0] + (pos[j, 0] - pos[i, 0]) + rng.normal(0, .3, n); r["a"]["y"][k, :n] = xy[:, 1] + (pos[j, 1] - pos[i, 1]) + rng.normal(0, .3, n)
        r["b"]["x"][k, :n] = xy[:, 0]; r["b"]["y"][k, :n] = xy[:, 1]
        r["a"]["id"][k, :n] = rng.integers(0, 2000, n); r["b"]["id"][k, :n] = rng.integers(0, 2000, n)
    r["accepted"][::13] = 0                                   # rejected pairs keep their inlier lists: they must not count
    r["n_in"][5::29] = 0
    return r


def check_alignment_equal(im, r, N, mom):
    fixed_sets = [None, [1 if k % 9 == 4 else 0 for k in range(N)], [1 if k >= N - 3 else 0 for k in range(N)]]
    labels = [None, np.array([0 if k % 11 == 5 else 1 for k in range(N)], np.int32)]
    assert np.array_equal(im.select_connected_moments(mom, N), im.select_connected_results(r, N))
    for fs in fixed_sets:
        for lab in labels:
            a = im.global_affine_align_results(r, N, fixed=fs, label=lab)
            b = im.global_affine_align_moments(mom, N, fixed=fs, label=lab)
            assert a.tobytes() == b.tobytes(), (fs is None, lab is None)


def test_alignment_from_host_moments_equals_alignment_from_records():
    import imagemosaicing_amd as im
    for seed, N, extra in ((3, 120, ()), (5, 37, ((9, 3), (20, 20)))):
        r = synthetic_records(im, seed, N, extra)
        mom = im.pair_moments_host(r)
        assert mom.dtype.itemsize == 184 and len(mom) == len(r)
        acc = (r["accepted"] != 0) & (r["n_in"] > 0)
        assert np.array_equal(mom["n_in"], np.where(acc, r["n_in"], 0)) and np.array_equal(mom["i"], r["i"]) and np.array_equal(mom["j"], r["j"])
        assert not mom["aa"][~acc].any() and not mom["ab"][~acc].any() and not mom["bb"][~acc].any()
        k = int(np.where(acc)[0][0]); n = int(r["n_in"][k])
        xa, ya = r["a"]["x"][k, :n].astype(np.float64), r["a"]["y"][k, :n].astype(np.float64)
        xb = r["b"]["x"][k, :n].astype(np.float64)
        assert mom["aa"][k][5] == n and mom["bb"][k][5] == n                                      # sum 1 * 1
        assert abs(mom["aa"][k][1] - (ya * xa).sum()) <= 1e-9 * abs((ya * xa).sum()) and abs(mom["ab"][k][0] - (xa * xb).sum()) <= 1e-9 * (xa * xb).sum()
        check_alignment_equal(im, r, N, mom)
        check_alignment_equal(im, r, N, mom[acc])           # what the exchange delivers: the accepted pairs only
    assert im.load_library().mi355_pair_moments_host(None, 3, None) < 0


@pytest.mark.gpu
def test_device_moments_equal_host_moments_bit_for_bit():
    import torch
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md
    ctx = im.Context(0)
    for seed, N in ((7, 150), (8, 20)):
        r = synthetic_records(im, seed, N)
        want = im.pair_moments_host(r)
        d_r = torch.from_numpy(r.view(np.uint8).reshape(len(r), -1).copy()).cuda()
        d_m = torch.full((len(r), im.PAIR_MOMENTS.itemsize), 0xCD, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.PairMomentsDev(d_r.data_ptr(), len(r), d_m.data_ptr()); ctx.synchronize()
        got = d_m.cpu().numpy().reshape(-1).view(im.PAIR_MOMENTS)
        assert got.tobytes() == want.tobytes()
        check_alignment_equal(im, r, N, got)
        # the exchange of a communicator of one rank, both transports: the accepted pairs' moments in record order
        acc = (r["accepted"] != 0)
        for transport in ("rccl", "torch"):
            ex = md.Exchange(ctx, transport, strict=True) if transport == "rccl" else md.Exchange(ctx, transport)
            m = ex.allgather_moments(d_r, len(r))
            ex.close()
            assert m.tobytes() == want[acc].tobytes(), transport
    ctx.close()


@pytest.mark.gpu
def test_alignment_of_a_real_survey_from_device_moments():
    """the same on the records a real pair stage leaves in HBM: a 13-frame survey, all pairs; labels and transforms from the device's moments
    equal the ones from the records, and the exchange's moments are those of the accepted records in record order"""
    import torch
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md
    from tests.synth_survey import render_frames
    ctx = im.Context(0)
    w, h, F = 800, 600, 13
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h, per_row=5)
    for k in range(F):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    pairs = im.pair_schedule(F, 182)
    res = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MatchPairsDev(pairs, res.data_ptr(), 2.5, 3); ctx.synchronize()
    r = res.cpu().numpy().reshape(-1).view(im.PAIR_RESULT)
    assert 8 <= int(r["accepted"].sum()) < len(pairs)
    ex = md.Exchange(ctx, "rccl", strict=True)
    mom = ex.allgather_moments(res, len(pairs))
    ex.close()
    assert mom.tobytes() == im.pair_moments_host(r)[r["accepted"] != 0].tobytes()
    label = im.select_connected_results(r, F)
    assert np.array_equal(label, im.select_connected_moments(mom, F)) and label.sum() >= 8
    fixed = [1 if (k == 0 or label[k] == 0) else 0 for k in range(F)]
    a = im.global_affine_align_results(r, F, fixed=fixed, label=label)
    b = im.global_affine_align_moments(mom, F, fixed=fixed, label=label)
    assert a.tobytes() == b.tobytes()
    ctx.close()
