"""CPU, world_size 2 over gloo: the N>1 path of the hot path -- pair sharding and the single all-gather of the
fixed-size pair records (the same code bench.py runs over RCCL)."""
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
