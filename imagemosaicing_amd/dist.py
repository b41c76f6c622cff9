"""Multi-GPU plumbing of the hot path: one process per GPU.

Sharding (the reference's own rule, MosaicWithoutPos.cpp:4861 / :5066, threads -> ranks): rank r extracts the frames
k mod G == r and matches the pairs (i, j) whose i it owns (mi355_pair_schedule); j runs over the reference's window
j in (i, i+182) (:5083-5084), so every rank needs every frame's features before matching.  Three exchanges, nothing else of the
data path crosses ranks:

  features   after detect+describe: all-gather of the fixed-size feature records (keypoints + u8 descriptors, 312 KB per
             frame) -- the reference hands features from the extraction threads to the matcher threads through
             d:/feature_temp files (:4874-4880 -> :5100-5103)
  results    after match + select + RANSAC: all-gather of the accepted pair records (H + inlier lists, 9664 B each) that
             feed Select_Connected_Matched_Images / global alignment -- PushMatchPairs under a mutex in the reference
             (:5236, :10137-10145) -- or, for surveys whose records weigh a gigabyte (C5), the records to ONE root rank (the one
             that runs the unchanged driver) and their 184-byte second moments to every rank for the replicated alignment
  frames     after the alignment: a rank holds only the frames it extracted; the frames its canvas stripe reads come from their
             owners (mi355_exchange_frames: ncclSend / ncclRecv groups) -- the reference composites in one address space
             (MosaicWithoutPos.cpp:4663 / :4671)

Transports: "rccl" = the C ABI's own collectives (mi355_allgather_features / mi355_allgather_results: ncclAllGather over
xGMI on the ctx stream; the product path, what bench.py runs with backend nccl); "torch" = the same packed records moved by
torch.distributed.all_gather (gloo in the CPU tests and in single-GPU dry runs, where RCCL cannot place two ranks on one
device).  Both go through the same pack / install entry points of the library.
"""
import sys

import numpy as np
import torch
import torch.distributed as dist

from .capi import PAIR_RESULT, FEATURE_HEADER, FEATURE_RECORD_BYTES, comm_unique_id, comm_available

REC = PAIR_RESULT.itemsize


def owned_frames(n_images, rank, world, rule="mod"):
    """frames rank `rank` extracts and holds.  "mod": k mod world == rank, the reference's own rule for its threads (MosaicWithoutPos.cpp:4861).
    "blocks": contiguous runs of ceil(n / world) frames -- the same load per rank, and frames that follow each other in a survey lie next to
    each other on the ground (flight lines), so a rank's frames fall into ONE band of the canvas: with the stripes dealt out by
    stripe_of_ranks() a stripe reads mostly its own rank's frames and mi355_exchange_frames moves the ones at the bands' edges only
    (C5, middle rank: 20 GB per step with "mod" ownership and box covers)."""
    if rule == "blocks":
        per = -(-n_images // world)
        return list(range(min(rank * per, n_images), min((rank + 1) * per, n_images)))
    return list(range(rank, n_images, world))


def frame_owner(n_images, world, rule="mod"):
    """owner[k] of every frame under `rule` (the `owner` argument of mi355_exchange_frames)"""
    if rule == "blocks":
        per = -(-n_images // world)
        return np.minimum(np.arange(n_images) // per, world - 1).astype(np.int32)
    return (np.arange(n_images) % world).astype(np.int32)


def exact_cover_pays(w, h, h9s, cw, ch, threshold=8.0):
    """whether a stripe's cover list should be formed exactly (mi355_mosaic_stripe_cover MI355_COVER_REFINED_EXACT: one device pass, ~0.5 ms
    per C3 stripe) or by box (host geometry, microseconds).  The exact list is shorter where many frames lie on top of each other: the
    measure is the mean number of frames per canvas pixel, sum of the frames' areas / canvas area -- C3's strip survey 3.0 (box 107 frames,
    exact 97: the pass costs more than the ten frames' 0.3 ms of wire), C5's block 59 (641 -> 340: 10 ms of wire for 1.5).  Replicated
    geometry: the same answer on every rank."""
    h9s = np.asarray(h9s, np.float64).reshape(-1, 9)
    valid = h9s[:, 8] != 0
    a = h9s[:, 0] * h9s[:, 4] - h9s[:, 1] * h9s[:, 3]            # area scale of the (near-affine) frame -> canvas maps
    area = float((np.abs(a) * np.asarray(w, np.float64) * np.asarray(h, np.float64))[valid].sum())
    return area / max(float(cw) * float(ch), 1.0) > threshold


def stripe_of_ranks(w, h, h9s, owner, world):
    """which canvas stripe each rank renders: the ranks in the order of the mean canvas row of the frames they hold (frame centres through the
    replicated transforms), so that a rank's stripe lies where its own frames are.  Pure host geometry on replicated data: the same answer on
    every rank.  Returns stripe[rank]."""
    h9s = np.asarray(h9s, np.float64).reshape(-1, 9)
    owner = np.asarray(owner)
    cx, cy = (np.asarray(w, np.float64) - 1) / 2, (np.asarray(h, np.float64) - 1) / 2
    den = h9s[:, 6] * cx + h9s[:, 7] * cy + h9s[:, 8]
    valid = h9s[:, 8] != 0
    y = np.where(valid, (h9s[:, 3] * cx + h9s[:, 4] * cy + h9s[:, 5]) / np.where(valid, den, 1.0), 0.0)
    mean_y = np.array([y[(owner == r) & valid].mean() if ((owner == r) & valid).any() else np.inf for r in range(world)])
    order = np.argsort(mean_y, kind="stable")          # order[s] = the rank that takes stripe s
    stripe = np.empty(world, np.int64)
    stripe[order] = np.arange(world)
    return stripe


def init_comm(ctx, group=None):
    """creates the ctx's RCCL communicator: rank 0 makes the 128-byte id, torch.distributed carries it to the other ranks.
    Raises on every rank alike when the library cannot reach librccl (rank 0 then broadcasts None instead of an id)."""
    if not dist.is_initialized():                      # single process: a communicator of one rank (exercises the same calls)
        ctx.CommInit(comm_unique_id(), 0, 1)
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        try:
            box = [comm_unique_id()]
        except Exception as e:                         # noqa: BLE001 -- reported below, on every rank
            box = [None]
            sys.stderr.write("mi355 dist: no RCCL unique id (%s)\n" % e)
    dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        raise RuntimeError("rank 0 could not create an RCCL unique id")
    ctx.CommInit(box[0], rank, world)


def _allgather_rows(local, n_max):
    """local: uint8 tensor [n_local, row_bytes] -> (uint8 [world, n_max, row_bytes], counts).  Ragged counts: the counts are
    gathered first and the payload is padded to n_max rows (None: the largest count)."""
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local.is_cuda:
        # gloo has no device collectives: stage through the host (CPU tests and single-GPU dry runs only)
        g, counts = _allgather_rows(local.cpu(), n_max)
        return g.to(local.device), counts
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1) if n_max is None else n_max
    row = local.shape[1]
    padded = local
    if local.shape[0] != n_max:
        padded = torch.zeros((n_max, row), dtype=torch.uint8, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * n_max, row), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    return out.view(world, n_max, row), counts


def compact_accepted(local):
    """keeps the records of accepted pairs only (what the reference pushes to the driver, MosaicWithoutPos.cpp:5201-5227): with
    the 182-frame pair window most scheduled pairs do not overlap, and the exchange shrinks by that factor (C4: 715 MB ->
    a few tens of MB per rank)"""
    off = PAIR_RESULT.fields["accepted"][1]
    acc = local[:, off:off + 4].contiguous().view(torch.int32).reshape(-1) != 0
    return local[acc]


def allgather_pair_results(local, n_local_max=None, accepted_only=False):
    """torch transport of the result exchange.  local: uint8 tensor [n_local, 9664] (device for nccl, cpu for gloo).
    Returns uint8 [world, n_max, 9664] and the per-rank counts."""
    if accepted_only:
        local = compact_accepted(local)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local.unsqueeze(0), [local.shape[0]]
    return _allgather_rows(local, n_local_max)


def gathered_to_records(gathered, counts):
    """uint8 [world, n_max, 9664] (+ counts) -> one structured numpy array of all valid records, rank-major"""
    g = gathered.cpu().numpy()
    parts = [g[r, :c].reshape(-1).view(PAIR_RESULT) for r, c in enumerate(counts)]
    return np.concatenate(parts) if parts else np.zeros(0, PAIR_RESULT)


def allgather_feature_records(hdr, payload, n_max=None):
    """torch transport of the feature exchange.  hdr: FEATURE_HEADER array [n_local]; payload: uint8 tensor
    [n_local, FEATURE_RECORD_BYTES].  Returns (headers [world][count_r], payload uint8 [world, n_max, REC], counts)."""
    h = torch.from_numpy(np.ascontiguousarray(hdr, FEATURE_HEADER).view(np.uint8).reshape(len(hdr), FEATURE_HEADER.itemsize).copy())
    if dist.get_backend() == "nccl":
        h = h.to(payload.device)                       # RCCL moves device tensors only
    gh, counts = _allgather_rows(h, n_max)
    gp, counts2 = _allgather_rows(payload, n_max)
    assert counts == counts2
    gh = gh.cpu()
    hdrs = [gh[r, :c].contiguous().numpy().reshape(-1).view(FEATURE_HEADER) for r, c in enumerate(counts)]
    return hdrs, gp, counts


def exchange_frames_torch(frames, h, ws, need, owner, rank, world):
    """torch.distributed transport of mi355_exchange_frames (gloo CPU tests, 2-rank dry runs on one device): the same table, walked in
    the same order by every rank -- owner sends, reader receives, frame after frame (a total order: no two ranks ever wait for each other
    crosswise).  frames: per frame a uint8 tensor (any device) where this rank holds it, else None; need: [G, n] table or this rank's own
    [n] row (then the rows are all-gathered first).  Returns (pointers, bytes received, bytes sent, {frame: received tensor})."""
    n = len(frames)
    need = np.ascontiguousarray(need, np.uint8)
    nccl = dist.is_initialized() and dist.get_backend() == "nccl"
    dev = next((f.device for f in frames if f is not None), torch.device("cpu"))
    if need.ndim == 1:                                 # this rank's own row: the rows of all ranks are gathered first
        if world > 1:
            rows = [torch.zeros(n, dtype=torch.uint8) for _ in range(world)]
            mine = torch.from_numpy(need.copy())
            if nccl:
                rows = [r.to(dev) for r in rows]; mine = mine.to(dev)
            dist.all_gather(rows, mine)
            need = np.stack([r.cpu().numpy() for r in rows])
        else:
            need = need[None, :]
    own = (lambda k: int(owner[k])) if owner is not None else (lambda k: k % world)
    recv = {}
    out, br, bs = [0] * n, 0, 0
    for k in range(n):
        o = own(k)
        nbytes = int(ws[k]) * int(h[k])
        if need[rank, k] and o == rank:
            out[k] = frames[k].data_ptr()
        for r in range(world):
            if not need[r, k] or r == o:
                continue
            if rank == o:
                t = frames[k].reshape(-1)[:nbytes]
                dist.send(t if nccl else t.cpu(), dst=r)
                bs += nbytes
            elif rank == r:
                buf = torch.empty(nbytes, dtype=torch.uint8, device=dev if nccl else "cpu")
                dist.recv(buf, src=o)
                buf = buf.to(dev)
                recv[k] = buf
                out[k] = buf.data_ptr()
                br += nbytes
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()  # the ctx stream may be another one: the copies must have landed
    return out, br, bs, recv


class Exchange:
    """The two exchanges of one rank's ctx.  transport "rccl": the library's own collectives on the ctx's communicator
    (init_comm); "torch": torch.distributed moves the same records.

    With "rccl" every rank first says whether it can bind librccl (mi355_comm_available touches no communicator) and the answers
    are MIN-reduced BEFORE any rank enters ncclCommInitRank -- a rank that cannot must not leave the others blocked in there.
    strict (the default): if any rank cannot, every rank raises; strict=False: every rank falls back to "torch" together, said
    loudly and visible in `self.transport`."""

    def __init__(self, ctx, transport="rccl", strict=True):
        assert transport in ("rccl", "torch")
        self.ctx, self.transport = ctx, transport
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rccl_ranks = None                         # what the communicator itself reports (ncclCommCount)
        self._payload = None
        self._recv_frames = {}                         # torch transport of exchange_frames: the received frames, alive until the next call
        if transport == "rccl":
            ok = 1 if comm_available() else 0
            if self.world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                all_ok = int(flag.item())
            else:
                all_ok = ok
            if not all_ok:
                msg = "rank %d: the C ABI cannot bind librccl on %s" % (self.rank, "this rank" if not ok else "another rank")
                if strict:
                    raise RuntimeError("mi355 dist: " + msg + " (transport 'rccl' requested; pass transport='torch' to move the records through torch.distributed)")
                sys.stderr.write("mi355 dist: " + msg + ": exchanges go through torch.distributed\n")
                self.transport = "torch"
            else:
                init_comm(ctx)                         # collective; raises on every rank alike if rank 0 has no id
                r, n = ctx.CommInfo()
                if (r, n) != (self.rank, self.world):
                    raise RuntimeError("mi355 dist: communicator reports rank %d of %d, expected %d of %d" % (r, n, self.rank, self.world))
                self.rccl_ranks = n

    def allgather_features(self, own_ids, n_max, device):
        """afterwards every frame of every rank is resident in this rank's ctx"""
        if self.transport == "rccl":
            self.ctx.AllGatherFeatures(own_ids, n_max)
            return
        if self.world == 1:
            return
        n = len(own_ids)
        if self._payload is None or self._payload.shape[0] < max(n, 1):
            self._payload = torch.empty((max(n, 1), FEATURE_RECORD_BYTES), dtype=torch.uint8, device=device)
        hdr = self.ctx.PackFeaturesDev(own_ids, self._payload.data_ptr()) if n else np.zeros(0, FEATURE_HEADER)
        hdrs, gp, counts = allgather_feature_records(hdr, self._payload[:n], n_max)
        if gp.is_cuda:
            torch.cuda.current_stream(gp.device).synchronize()     # the ctx stream may be another one: the records must have landed
        for r in range(self.world):
            if r == self.rank or counts[r] == 0:
                continue
            block = gp[r].contiguous()
            self.ctx.InstallFeaturesDev(hdrs[r], block.data_ptr())

    def allgather_results(self, results, n_local, accepted_only=True, root=-1, copy=True, wait=True):
        """results: uint8 device tensor [>= n_local, 9664] written by MatchPairsDev.  root < 0: returns all ranks' records (numpy, rank-major)
        on every rank; root >= 0: on that rank only (the others send and get an empty array) -- the rank that runs the reference's unchanged
        driver on the inlier lists (MosaicWithoutPos.cpp:4575-4591).  copy=False: a view of the library's pinned buffer, valid until the next call."""
        if self.transport == "rccl":
            return self.ctx.AllGatherResults(results.data_ptr(), n_local, accepted_only, root=root, copy=copy, wait=wait)
        g, counts = allgather_pair_results(results[:n_local], accepted_only=accepted_only)
        if root >= 0 and self.rank != root:
            return np.zeros(0, PAIR_RESULT)
        return gathered_to_records(g, counts)

    def stripe_need(self, w, h, h9s, stripes, blended=False, keep=None, band=5, exact=False):
        """the G x n table of mi355_exchange_frames: row r = the frames rank r's stripe (row0, rows) reads; the same on every rank"""
        return np.stack([self.ctx.StripeCover(w, h, h9s, r0, nr, blended=blended, keep=keep, band=band, exact=exact) for (r0, nr) in stripes])

    def exchange_frames(self, frames, h, ws, need, owner=None, own_through_rccl=False):
        """frames: per frame a torch uint8 device tensor (rccl transport: or its device address as an int) where this rank holds it, else None.  Every frame this rank's stripe reads
        (need[rank]) and does not hold comes from its owner (k mod G unless `owner` says otherwise).  Returns (device pointers for the
        stripe calls, 0 where the stripe does not read the frame; bytes received; bytes sent); received frames stay alive until the next call."""
        n = len(frames)
        need = np.ascontiguousarray(need, np.uint8)
        if self.transport == "rccl":
            return self.ctx.ExchangeFrames([(f if isinstance(f, int) else f.data_ptr()) if f is not None else 0 for f in frames], h, ws, need, owner=owner, own_through_rccl=own_through_rccl)
        out, br, bs, self._recv_frames = exchange_frames_torch(frames, h, ws, need, owner, self.rank, self.world)
        return out, br, bs

    def allgather_moments(self, results, n_local, copy=True):
        """what the alignment needs of this rank's accepted pairs (PAIR_MOMENTS, 184 B per pair, formed on the device) from every rank (numpy, rank-major)"""
        if self.transport == "rccl":
            return self.ctx.AllGatherMoments(results.data_ptr(), n_local, copy=copy)
        from .capi import PAIR_MOMENTS
        local = compact_accepted(results[:n_local]).contiguous()
        mom = torch.zeros((max(local.shape[0], 1), PAIR_MOMENTS.itemsize), dtype=torch.uint8, device=local.device)
        if local.shape[0] > 0:
            torch.cuda.current_stream(local.device).synchronize()
            self.ctx.PairMomentsDev(local.data_ptr(), local.shape[0], mom.data_ptr())
            self.ctx.synchronize()
        mom = mom[:local.shape[0]]
        if dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() != "nccl":
                mom = mom.cpu()
            g, counts = _allgather_rows(mom, None)
        else:
            g, counts = mom.unsqueeze(0), [mom.shape[0]]
        g = g.cpu().numpy()
        parts = [g[r, :c].reshape(-1).view(PAIR_MOMENTS) for r, c in enumerate(counts)]
        return np.concatenate(parts) if parts else np.zeros(0, PAIR_MOMENTS)

    def close(self):
        if self.transport == "rccl":
            self.ctx.CommDestroy()
