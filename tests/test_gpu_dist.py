"""GPU: the N>1 path on real kernels.

  * world 2 on ONE device over gloo ("torch" transport: RCCL cannot place two ranks on one GPU): ONE survey, frames k mod 2 and
    pairs i mod 2 per rank, feature records exchanged, results gathered -- the union equals what a single rank computes
    alone, record for record, bit for bit (C4-mini: window 182 => all pairs).
  * world 1 through the C ABI's own RCCL collectives (mi355_comm_init / mi355_allgather_features / mi355_allgather_results):
    the calls the 8-GPU run makes, on a communicator of one rank.
"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md
    from tests.synth_survey import render_frames

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = im.Context(0)
    w, h, F, window = 800, 600, 13, 182
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h, per_row=5)      # the same survey on every rank
    own = md.owned_frames(F, rank, world)
    pairs = im.pair_schedule(F, window, rank, world)
    assert (pairs[:, 0] %% world == rank).all()
    ex = md.Exchange(ctx, "torch")
    for k in own:
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    ex.allgather_features(own, (F + world - 1) // world, "cuda")
    # a frame owned by the other rank is resident now, with the same bytes the owner holds
    other = (rank + 1) %% world
    kp_o, d_o = ctx.GetFeatures(other)
    results = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")
    ctx.MatchPairsDev(pairs, results.data_ptr(), 2.5, 9)
    acc = ex.allgather_results(results, len(pairs), accepted_only=True)
    full = ex.allgather_results(results, len(pairs), accepted_only=False)
    acc = acc[np.lexsort((acc["j"], acc["i"]))]
    full = full[np.lexsort((full["j"], full["i"]))]
    if rank == 0:
        # the whole survey on one rank
        c1 = im.Context(0)
        for k in range(F):
            c1.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
        allp = im.pair_schedule(F, window)
        ref = c1.MatchPairs(allp, 2.5, 9)
        kp_r, d_r = c1.GetFeatures(other)
        assert np.array_equal(kp_o.view(np.uint8), kp_r.view(np.uint8)) and np.array_equal(d_o, d_r), "installed features differ from the owner's"
        assert len(full) == len(ref) == F * (F - 1) // 2
        assert np.array_equal(full.view(np.uint8), ref.view(np.uint8)), "union of the ranks' records differs from the single-rank records"
        racc = ref[ref["accepted"] == 1]
        assert len(acc) == len(racc) and np.array_equal(acc.view(np.uint8), racc.view(np.uint8))
        assert 0 < len(racc) < len(ref)
        # the driver step on the gathered records gives the same transforms on every rank count
        T1 = im.global_affine_align(im.results_to_match_pairs(racc), F)
        T2 = im.global_affine_align(im.results_to_match_pairs(acc), F)
        assert np.array_equal(T1["m"].view(np.uint32), T2["m"].view(np.uint32))
        c1.close()
        print("DIST_GPU_OK", len(ref), len(racc))
    # ---- results to ONE root (the rank that runs the unchanged driver), moments to every rank -------------------------------------------------
    root_only = ex.allgather_results(results, len(pairs), accepted_only=True, root=1)
    if rank == 1:
        root_only = root_only[np.lexsort((root_only["j"], root_only["i"]))]
        assert np.array_equal(root_only.view(np.uint8), acc.view(np.uint8)), "records gathered on the root differ from the all-gathered ones"
    else:
        assert len(root_only) == 0
    # ---- frame ownership (SURVEY 8e primary form): a rank holds the frames it extracted; its stripes read the others' through the exchange ----
    from tests.synth_survey import affine3
    Hgt = np.stack([(np.linalg.inv(affine3(A[0])) @ affine3(A[k])).reshape(9) for k in range(F)]).astype(np.float32)
    wv, hv, wsv = [w] * F, [h] * F, [ws] * F
    held = [frames[k] if k %% world == rank else None for k in range(F)]
    full_ptr = [frames[k].data_ptr() for k in range(F)]
    # (a) MosaicImagesRefined stripes
    cw, ch, cws, _ = im.mosaic_layout(wv, hv, Hgt)
    stripes = [((ch * r) // world, (ch * (r + 1)) // world - (ch * r) // world) for r in range(world)]
    need = ex.stripe_need(wv, hv, Hgt, stripes)
    ptrs, br, bs = ex.exchange_frames(held, hv, wsv, need)
    assert br > 0 and bs > 0, (br, bs)
    assert all((p != 0) == bool(need[rank, k]) for k, p in enumerate(ptrs))
    row0, rows = stripes[rank]
    a = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda"); b = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MosaicImagesRefinedDev(ptrs, wv, hv, wsv, Hgt, a.data_ptr(), cw, ch, cws, row0, rows)
    ctx.MosaicImagesRefinedDev(full_ptr, wv, hv, wsv, Hgt, b.data_ptr(), cw, ch, cws, row0, rows)
    ctx.synchronize()
    assert torch.equal(a, b) and int(a.count_nonzero()) > 0, "refined stripe from owner-only frames + exchange differs from the replicas' stripe"
    # (received copies live until the next exchange: the stripe above was rendered first)  The same with frames owned in blocks, the stripes dealt out to where a rank's frames are, and every rank contributing only ITS exact row
    owner_b = md.frame_owner(F, world, "blocks")
    held_b = [frames[k] if owner_b[k] == rank else None for k in range(F)]
    sidx = md.stripe_of_ranks(wv, hv, Hgt, owner_b, world)
    assert sorted(sidx.tolist()) == list(range(world))
    r0b, nrb = stripes[int(sidx[rank])]
    row_b = ctx.StripeCover(wv, hv, Hgt, r0b, nrb, exact=True)
    pb, brb, bsb = ex.exchange_frames(held_b, hv, wsv, row_b, owner=owner_b)
    assert all((p != 0) == bool(row_b[k]) for k, p in enumerate(pb))
    ab = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda"); bb_ = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MosaicImagesRefinedDev(pb, wv, hv, wsv, Hgt, ab.data_ptr(), cw, ch, cws, r0b, nrb)
    ctx.MosaicImagesRefinedDev(full_ptr, wv, hv, wsv, Hgt, bb_.data_ptr(), cw, ch, cws, r0b, nrb)
    ctx.synchronize()
    assert torch.equal(ab, bb_), "stripe from block-owned frames + exact cover rows differs from the replicas' stripe"
    tb = torch.tensor([brb, bsb], dtype=torch.int64); dist.all_reduce(tb)
    assert int(tb[0]) == int(tb[1])
    # (b) LaplacianPyramidBlending stripes (chips that reach the rows + the pyramids' reach)
    keep = im.resample_by_overlap(wv, hv, Hgt, 0.7)
    bw_, bh_, _ = im.blend_layout(wv, hv, Hgt, keep)
    bstripes = [((bh_ * r) // world, (bh_ * (r + 1)) // world - (bh_ * r) // world) for r in range(world)]
    bneed = ex.stripe_need(wv, hv, Hgt, bstripes, blended=True, keep=keep, band=5)
    assert all(bneed[rank, k] == 0 for k in range(F) if not keep[k])
    bptrs, bbr, bbs = ex.exchange_frames(held, hv, wsv, bneed)
    br0, brows = bstripes[rank]
    o1, _, _, _ = ctx.MosaicBlendedDev(bptrs, wv, hv, wsv, Hgt, keep=keep, band=5, row0=br0, rows=brows)
    o2, _, _, _ = ctx.MosaicBlendedDev(full_ptr, wv, hv, wsv, Hgt, keep=keep, band=5, row0=br0, rows=brows)
    assert torch.equal(o1, o2) and int(o1.count_nonzero()) > 0, "blended stripe from owner-only frames + exchange differs from the replicas' stripe"
    tot = torch.tensor([br + bbr, bs + bbs], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot[0]) == int(tot[1])                     # every byte sent was received
    if rank == 0:
        print("FRAME_EXCHANGE_OK", int(tot[0]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_one_survey_equals_single_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST_GPU_OK" in r.stdout and "FRAME_EXCHANGE_OK" in r.stdout


def test_rccl_collectives_world1():
    import torch
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md
    from tests.synth_survey import render_frames
    ctx = im.Context(0)
    w, h, F = 640, 480, 5
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h, per_row=5)
    assert im.comm_available()
    ex = md.Exchange(ctx, "rccl")                      # mi355_comm_unique_id + mi355_comm_init(rank 0 of 1)
    assert ex.rccl_ranks == 1 and ctx.CommInfo() == (0, 1)        # what ncclCommUserRank / ncclCommCount report
    for k in range(F):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    before = [ctx.GetFeatures(k) for k in range(F)]
    ctx.AllGatherFeatures(list(range(F)), F)           # ncclAllGather of headers + records; own frames stay as they are
    ctx.AllGatherFeatures([0, 2, 4], F)                # fewer frames than n_max: padding records are skipped
    # a rank with a bad local argument still takes part in the collectives and then reports the error
    with pytest.raises(im.Mi355Error):
        ctx.AllGatherFeatures([0, 77], F)              # image 77 has no features
    ctx.AllGatherFeatures([1, 3], F)                   # the communicator is still usable afterwards
    for k in range(F):
        kp, d = ctx.GetFeatures(k)
        assert np.array_equal(kp.view(np.uint8), before[k][0].view(np.uint8)) and np.array_equal(d, before[k][1])
    pairs = im.pair_schedule(F, 182)
    results = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")
    ctx.MatchPairsDev(pairs, results.data_ptr(), 2.5, 4)
    ctx.synchronize()
    loc = results.cpu().numpy().reshape(-1).view(im.PAIR_RESULT)
    allr = ex.allgather_results(results, len(pairs), accepted_only=False)
    assert np.array_equal(allr.view(np.uint8), loc.view(np.uint8))
    acc = ex.allgather_results(results, len(pairs), accepted_only=True)
    want = loc[loc["accepted"] == 1]
    assert len(acc) == len(want) > 0 and np.array_equal(acc.view(np.uint8), want.view(np.uint8))
    # pack -> install round trip under another id space (the transport-agnostic halves)
    payload = torch.empty((2, im.FEATURE_RECORD_BYTES), dtype=torch.uint8, device="cuda")
    hdr = ctx.PackFeaturesDev([1, 3], payload.data_ptr())
    assert hdr["n_kp"].tolist() == [len(before[1][0]), len(before[3][0])]
    hdr["img_id"] = [101, 103]
    ctx.InstallFeaturesDev(hdr, payload.data_ptr())
    for a, b in ((101, 1), (103, 3)):
        kp, d = ctx.GetFeatures(a)
        assert np.array_equal(kp.view(np.uint8), before[b][0].view(np.uint8)) and np.array_equal(d, before[b][1])
    r1 = ctx.MatchPairs([(101, 103), (1, 3)], 2.5, 4)
    assert np.array_equal(r1[0]["a"].view(np.uint8), r1[1]["a"].view(np.uint8)) and np.array_equal(r1[0]["H"].view(np.uint32), r1[1]["H"].view(np.uint32))
    ex.close()
    ctx.close()


def test_compaction_of_accepted_records_keeps_order_at_every_size():
    """mi355_compact_accepted_dev (the first step of mi355_allgather_results): the accepted records to the front, order kept
    (MosaicWithoutPos.cpp:5201-5227 pushes them in loop order).  Three launches -- accepted per block of 256, a scan of the block counts by one
    workgroup in chunks of 1024 blocks, the moves -- so: sizes around the block and the chunk, none / all / sparse / dense accepted."""
    import torch
    import imagemosaicing_amd as im
    ctx = im.Context(0)
    rng = np.random.default_rng(5)
    rec = im.PAIR_RESULT.itemsize
    for n, p in [(1, 1.0), (1, 0.0), (255, 0.5), (256, 0.5), (257, 1.0), (5000, 0.03), (74029, 0.03), (262144 + 300, 0.4), (300000, 0.0), (270000, 1.0)]:
        host = np.zeros(n, im.PAIR_RESULT)
        host["accepted"] = (rng.random(n) < p).astype(host["accepted"].dtype)
        host["i"] = np.arange(n); host["j"] = rng.integers(0, 1 << 20, n); host["n_in"] = rng.integers(0, 400, n)
        host["H"] = rng.random((n, 9)).astype(np.float32)
        raw = host.view(np.uint8).reshape(n, rec)
        raw[:, -16:] = rng.integers(0, 256, (n, 16), dtype=np.uint8)          # the record's last 16-byte piece moves too
        host = raw.view(im.PAIR_RESULT).reshape(n)
        d_in = torch.from_numpy(raw.copy()).cuda()
        d_out = torch.full((n, rec), 0xAB, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()                                              # (the fill runs on torch's stream, the library on its own)
        k = ctx.CompactAcceptedDev(d_in.data_ptr(), n, d_out.data_ptr())
        want = raw[host["accepted"] != 0]
        assert k == len(want), (n, p, k, len(want))
        got = d_out.cpu().numpy()
        if not np.array_equal(got[:k], want):
            bad = np.where((got[:k] != want).any(1))[0]
            raise AssertionError((n, p, len(bad), bad[:5].tolist(), bad[-3:].tolist(), [int(x) for x in np.where(got[bad[0]] != want[bad[0]])[0][:6]]))
        assert (got[k:] == 0xAB).all(), (n, p)                               # nothing written behind the accepted records
    ctx.close()


def test_stripe_cover_is_exactly_what_the_stripe_calls_read():
    """mi355_mosaic_stripe_cover (the table mi355_exchange_frames is driven by): the stripe calls succeed and give the whole canvas's rows
    with every frame OUTSIDE the cover withheld (pointer 0), and fail when a frame INSIDE the (exact) cover is withheld -- for the last-write-wins canvas
    (MosaicWithoutPos.cpp:2194) and for the blended one (MosaicImage.cpp:2205), at several cuts"""
    import torch
    import imagemosaicing_amd as im
    from tests.synth_survey import render_frames, affine3
    ctx = im.Context(0)
    w, h, F = 640, 480, 12
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h, per_row=4)
    H = np.stack([(np.linalg.inv(affine3(A[0])) @ affine3(A[k])).reshape(9) for k in range(F)]).astype(np.float32)
    H[5, 8] = 0.0                                                      # an invalid image (h.m[8] == 0) is never read
    wv, hv, wsv = [w] * F, [h] * F, [ws] * F
    full = [frames[k].data_ptr() for k in range(F)]
    cw, ch, cws, _ = im.mosaic_layout(wv, hv, H)
    whole = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MosaicImagesRefinedDev(full, wv, hv, wsv, H, whole.data_ptr(), cw, ch, cws)
    ctx.synchronize()
    whole = whole.reshape(ch, cws)
    keep = im.resample_by_overlap(wv, hv, H, 0.7)
    bwhole, bw_, bh_, bws_ = ctx.MosaicBlendedDev(full, wv, hv, wsv, H, keep=keep, band=5)
    some_partial = some_exact_smaller = False
    for G in (2, 3, 5):
        for r in range(G):
            row0, rows = (ch * r) // G, (ch * (r + 1)) // G - (ch * r) // G
            need = ctx.StripeCover(wv, hv, H, row0, rows)
            assert need[5] == 0
            some_partial |= bool(0 < need.sum() < F - 1)
            ptrs = [full[k] if need[k] else 0 for k in range(F)]
            out = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.MosaicImagesRefinedDev(ptrs, wv, hv, wsv, H, out.data_ptr(), cw, ch, cws, row0, rows)
            ctx.synchronize()
            assert torch.equal(out.reshape(ch, cws)[row0:row0 + rows], whole[row0:row0 + rows])
            # the exact cover (the frames that GIVE a pixel of these rows its sample: the tile kernel's walk without its loads) is a subset of the
            # box cover, and the stripe needs nothing else
            exact = ctx.StripeCover(wv, hv, H, row0, rows, exact=True)
            assert (exact <= need).all() and exact.sum() > 0
            some_exact_smaller |= bool(exact.sum() < need.sum())
            pe = [full[k] if exact[k] else 0 for k in range(F)]
            out2 = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.MosaicImagesRefinedDev(pe, wv, hv, wsv, H, out2.data_ptr(), cw, ch, cws, row0, rows)
            ctx.synchronize()
            assert torch.equal(out2.reshape(ch, cws)[row0:row0 + rows], whole[row0:row0 + rows])
            pe[int(np.flatnonzero(exact)[-1])] = 0                      # ... and nothing less: with the check switched on the call names the frame it misses
            ctx.set_option("strict_frames", 1)
            with pytest.raises(im.Mi355Error):
                ctx.MosaicImagesRefinedDev(pe, wv, hv, wsv, H, out2.data_ptr(), cw, ch, cws, row0, rows)
            ctx.set_option("strict_frames", 0)
            # blended
            b0, brows = (bh_ * r) // G, (bh_ * (r + 1)) // G - (bh_ * r) // G
            bneed = ctx.StripeCover(wv, hv, H, b0, brows, blended=True, keep=keep, band=5)
            assert bneed[5] == 0 and all(bneed[k] == 0 for k in range(F) if not keep[k])
            bp = [full[k] if bneed[k] else 0 for k in range(F)]
            o, _, _, _ = ctx.MosaicBlendedDev(bp, wv, hv, wsv, H, keep=keep, band=5, row0=b0, rows=brows)
            assert torch.equal(o, bwhole[b0:b0 + brows])
            bp[int(np.flatnonzero(bneed)[-1])] = 0
            with pytest.raises(im.Mi355Error):
                ctx.MosaicBlendedDev(bp, wv, hv, wsv, H, keep=keep, band=5, row0=b0, rows=brows)
    assert some_partial and some_exact_smaller                          # the cuts really left frames out, and the exact lists are shorter than the box lists somewhere
    ctx.close()


def test_owner_only_frames_plus_exchange_equal_replicas_rccl_world1():
    """mi355_exchange_frames on the C ABI's own communicator (one rank: RCCL cannot place two on one device): with
    MI355_EXCHANGE_OWN_THROUGH_RCCL every frame the stripe reads goes through ncclSend / ncclRecv (to itself) into the ctx's landing area,
    in groups, and the stripes rendered from the RECEIVED copies are the replicas' stripes byte for byte -- both canvases.  Also: a root
    gather of the pair records (root = 0 of 1) equals the all-gather, and the library's pinned result buffer is reused, not reallocated."""
    import torch
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md
    from tests.synth_survey import render_frames, affine3
    ctx = im.Context(0)
    w, h, F = 640, 480, 70                                               # more than one group of 64 frames
    frames, A, gains, ws = render_frames(ctx, torch, F, w, h, per_row=10)
    H = np.stack([(np.linalg.inv(affine3(A[0])) @ affine3(A[k])).reshape(9) for k in range(F)]).astype(np.float32)
    wv, hv, wsv = [w] * F, [h] * F, [ws] * F
    full = [frames[k].data_ptr() for k in range(F)]
    ex = md.Exchange(ctx, "rccl")
    cw, ch, cws, _ = im.mosaic_layout(wv, hv, H)
    row0, rows = ch // 3, ch // 2
    need = ex.stripe_need(wv, hv, H, [(row0, rows)])
    assert 0 < need.sum() < F
    ptrs, br, bs = ex.exchange_frames([frames[k] for k in range(F)], hv, wsv, need, own_through_rccl=True)
    assert br == 0 and bs == 0                                           # nothing crossed ranks ...
    assert all((p != 0) == bool(need[0, k]) for k, p in enumerate(ptrs))
    assert all(p != full[k] for k, p in enumerate(ptrs) if p)            # ... but every frame read is the copy ncclRecv delivered
    a = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda"); b = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MosaicImagesRefinedDev(ptrs, wv, hv, wsv, H, a.data_ptr(), cw, ch, cws, row0, rows)
    ctx.MosaicImagesRefinedDev(full, wv, hv, wsv, H, b.data_ptr(), cw, ch, cws, row0, rows)
    ctx.synchronize()
    assert torch.equal(a, b) and int(a.count_nonzero()) > 0
    keep = im.resample_by_overlap(wv, hv, H, 0.7)
    bw_, bh_, _ = im.blend_layout(wv, hv, H, keep)
    b0, brows = bh_ // 4, bh_ // 3
    bneed = ex.stripe_need(wv, hv, H, [(b0, brows)], blended=True, keep=keep, band=5)
    bp, _, _ = ex.exchange_frames([frames[k] for k in range(F)], hv, wsv, bneed, own_through_rccl=True)
    o1, _, _, _ = ctx.MosaicBlendedDev(bp, wv, hv, wsv, H, keep=keep, band=5, row0=b0, rows=brows)
    o2, _, _, _ = ctx.MosaicBlendedDev(full, wv, hv, wsv, H, keep=keep, band=5, row0=b0, rows=brows)
    assert torch.equal(o1, o2) and int(o1.count_nonzero()) > 0
    # the rank's own row alone (exact cover from its own device): the rows are all-gathered inside the call (n bytes per rank)
    exact = ctx.StripeCover(wv, hv, H, row0, rows, exact=True)
    assert (exact <= need[0]).all()
    pl, _, _ = ex.exchange_frames([frames[k] for k in range(F)], hv, wsv, exact, own_through_rccl=True)
    assert all((p != 0) == bool(exact[k]) for k, p in enumerate(pl))
    a2 = torch.zeros(ch * cws, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.MosaicImagesRefinedDev(pl, wv, hv, wsv, H, a2.data_ptr(), cw, ch, cws, row0, rows)
    ctx.synchronize()
    assert torch.equal(a2, b)
    # without the flag the rank's own frames are handed back as they are
    p2, _, _ = ex.exchange_frames([frames[k] for k in range(F)], hv, wsv, need)
    assert all(p == (full[k] if need[0, k] else 0) for k, p in enumerate(p2))
    # a rank that owns a frame it cannot produce fails before any transfer is posted
    with pytest.raises(im.Mi355Error):
        ex.exchange_frames([None] * F, hv, wsv, need)
    # pair records: root gather == all-gather; the result lives in the ctx's pinned buffer (same address on the second call)
    for k in range(8):
        ctx.SiftExtractDev(k, frames[k].data_ptr(), w, h, ws)
    pairs = im.pair_schedule(8, 182)
    results = torch.zeros((len(pairs), im.PAIR_RESULT.itemsize), dtype=torch.uint8, device="cuda")
    ctx.MatchPairsDev(pairs, results.data_ptr(), 2.5, 4)
    ctx.synchronize()
    r_all = ex.allgather_results(results, len(pairs), accepted_only=True)
    v1 = ex.allgather_results(results, len(pairs), accepted_only=True, root=0, copy=False)
    addr1 = v1.ctypes.data
    assert len(r_all) > 0 and np.array_equal(v1.view(np.uint8), r_all.view(np.uint8))
    v2 = ex.allgather_results(results, len(pairs), accepted_only=True, root=-1, copy=False)
    assert v2.ctypes.data == addr1 and np.array_equal(v2.view(np.uint8), r_all.view(np.uint8))
    with pytest.raises(im.Mi355Error):
        ex.allgather_results(results, len(pairs), accepted_only=True, root=1)      # no such rank
    m1 = ex.allgather_moments(results, len(pairs))
    assert len(m1) == len(r_all) and np.array_equal(m1["i"], r_all["i"]) and np.array_equal(m1["n_in"], r_all["n_in"])
    ex.close()
    ctx.close()
