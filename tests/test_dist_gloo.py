"""CPU, world_size 2 over gloo: the N>1 path of the hot path -- frame / pair sharding, the two all-gathers (feature records,
pair records) and the frame exchange (owners send the frames a stripe reads) with the "torch" transport of imagemosaicing_amd/dist.py (the same records the C ABI moves over RCCL; the
kernels around them are covered on the GPU by tests/test_gpu_dist.py)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import imagemosaicing_amd as im
    from imagemosaicing_amd import dist as md

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, window = 37, 5
    pairs = im.pair_schedule(N, window, rank, world)
    assert (pairs[:, 0] %% world == rank).all()
    # fabricate this rank's results: a translation-only survey, 40 "inliers" per pair
    rec = np.zeros(len(pairs), im.PAIR_RESULT)
    rec["i"] = pairs[:, 0]; rec["j"] = pairs[:, 1]; rec["n_in"] = 40; rec["accepted"] = 1; rec["ok"] = 1
    rng = np.random.default_rng(rank)
    for k, (i, j) in enumerate(pairs):
        xy = rng.uniform(0, 100, (40, 2)).astype(np.float32)
        rec["a"]["x"][k, :40] = xy[:, 0] + 10.0 * (j - i); rec["a"]["y"][k, :40] = xy[:, 1]
        rec["b"]["x"][k, :40] = xy[:, 0];                   rec["b"]["y"][k, :40] = xy[:, 1]
    local = torch.from_numpy(rec.view(np.uint8).reshape(len(pairs), -1).copy())
    gathered, counts = md.allgather_pair_results(local)
    allrec = md.gathered_to_records(gathered, counts)
    want = im.pair_schedule(N, window)
    assert sum(counts) == len(want) == len(allrec)
    assert {(int(a), int(b)) for a, b in zip(allrec["i"], allrec["j"])} == {(int(a), int(b)) for a, b in want}
    # every rank can now run the driver step: connected component + global affine alignment
    mp = im.results_to_match_pairs(allrec)
    label = im.select_connected(mp, N)
    assert label.sum() == N
    T = im.global_affine_align(mp, N)
    tx = T["m"][:, 2]
    assert np.abs(tx - 10.0 * np.arange(N)).max() < 1e-2, tx[:5]
    # accepted-only exchange: rejected pairs never leave their rank
    rec2 = rec.copy()
    rec2["accepted"][::3] = 0
    local2 = torch.from_numpy(rec2.view(np.uint8).reshape(len(pairs), -1).copy())
    g2, c2 = md.allgather_pair_results(local2, accepted_only=True)
    all2 = md.gathered_to_records(g2, c2)
    assert (all2["accepted"] == 1).all() and c2[rank] == int(rec2["accepted"].sum()) and len(all2) == sum(c2) < len(want)
    # feature exchange, torch transport: ragged per-rank frame counts (19 frames over 2 ranks: 10 + 9), padded to n_max
    F = 19
    own = md.owned_frames(F, rank, world)
    assert own == list(range(rank, F, world))
    hdr = np.zeros(len(own), im.FEATURE_HEADER)
    hdr["img_id"] = own; hdr["n_kp"] = [100 + k for k in own]; hdr["w"] = 640; hdr["h"] = 480
    payload = torch.empty((len(own), im.FEATURE_RECORD_BYTES), dtype=torch.uint8)
    for q, k in enumerate(own):
        payload[q] = k %% 251
    hdrs, gp, cnts = md.allgather_feature_records(hdr, payload, (F + world - 1) // world)
    assert cnts == [10, 9] and gp.shape == (2, 10, im.FEATURE_RECORD_BYTES)
    seen = []
    for r in range(world):
        assert hdrs[r]["img_id"].tolist() == list(range(r, F, world))
        for q, k in enumerate(hdrs[r]["img_id"]):
            assert int(hdrs[r]["n_kp"][q]) == 100 + k and bool((gp[r, q] == k %% 251).all())
            seen.append(int(k))
    assert sorted(seen) == list(range(F))
    # every pair of the reference's window is owned by exactly one rank, and its i-frame by the same rank
    for win in (2, 5, 182):
        allp = {(int(a), int(b)) for a, b in im.pair_schedule(F, win)}
        mine = {(int(a), int(b)) for a, b in im.pair_schedule(F, win, rank, world)}
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([len(mine)], dtype=torch.int64))
        assert mine <= allp and sum(int(x) for x in sizes) == len(allp) and all(a %% world == rank for a, b in mine)
    # ---- frame ownership + exchange (SURVEY 8e primary form), torch transport of mi355_exchange_frames on CPU tensors ----------------------
    Fx, hx, wsx = 11, 6, 16
    for rule in ("mod", "blocks"):
        owner = md.frame_owner(Fx, world, rule)
        mine = md.owned_frames(Fx, rank, world, rule)
        assert mine == [k for k in range(Fx) if owner[k] == rank] and sorted(set(owner.tolist())) == list(range(world))
        held = [torch.full((hx * wsx,), 10 * k + 1, dtype=torch.uint8) if owner[k] == rank else None for k in range(Fx)]
        rng2 = np.random.default_rng(5)                           # the same table on every rank
        need = (rng2.random((world, Fx)) < 0.6).astype(np.uint8)
        need[:, 3] = 0                                            # a frame nobody reads
        for local_rows in (False, True):
            ptrs, br, bs, recv = md.exchange_frames_torch(held, [hx] * Fx, [wsx] * Fx, need[rank] if local_rows else need, owner, rank, world)
            for k in range(Fx):
                if not need[rank, k]:
                    assert ptrs[k] == 0 and k not in recv
                elif owner[k] == rank:
                    assert ptrs[k] == held[k].data_ptr() and k not in recv
                else:
                    assert ptrs[k] == recv[k].data_ptr() and bool((recv[k] == 10 * k + 1).all()) and recv[k].numel() == hx * wsx
            assert br == hx * wsx * sum(1 for k in range(Fx) if need[rank, k] and owner[k] != rank)
            assert bs == hx * wsx * sum(int(need[r, k]) for r in range(world) for k in mine if r != rank)
            tot = torch.tensor([br, bs], dtype=torch.int64); dist.all_reduce(tot)
            assert int(tot[0]) == int(tot[1])
    # the stripes are dealt out to where a rank's frames lie: frames whose canvas row falls with the index -> rank 0 (low indices) gets the LAST stripe
    w9 = [100] * Fx; h9 = [80] * Fx
    Hs = np.tile(np.eye(3, dtype=np.float32).reshape(9), (Fx, 1)); Hs[:, 5] = 1000.0 - 90.0 * np.arange(Fx)
    sidx = md.stripe_of_ranks(w9, h9, Hs, md.frame_owner(Fx, world, "blocks"), world)
    assert sidx.tolist() == [1, 0]
    Hs[:, 5] = 90.0 * np.arange(Fx)
    assert md.stripe_of_ranks(w9, h9, Hs, md.frame_owner(Fx, world, "blocks"), world).tolist() == [0, 1]
    if rank == 0:
        print("GLOO_OK", counts, c2)
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_pair_sharding_and_allgather_world2(tmp_path):
    from imagemosaicing_amd import build
    build.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "GLOO_OK" in r.stdout
