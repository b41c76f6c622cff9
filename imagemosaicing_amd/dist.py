"""Multi-GPU plumbing of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards embarrassingly: rank r owns frames and pairs of its own block (mi355_pair_schedule strides the
reference's i-loop by rank exactly like its threads, MosaicWithoutPos.cpp:5066).  There is no data-path
collective; the ONLY exchange is one all-gather of the fixed-size per-pair result records (H + inlier lists,
9664 B each) that feed global alignment -- what the reference's threads do through PushMatchPairs under a mutex
(MosaicWithoutPos.cpp:5236, 10137-10145).
"""
import numpy as np
import torch
import torch.distributed as dist

from .capi import PAIR_RESULT

REC = PAIR_RESULT.itemsize


def compact_accepted(local):
    """keeps the records of accepted pairs only (what the reference pushes to the driver, MosaicWithoutPos.cpp:5201-5227): with
    the 182-frame pair window most scheduled pairs do not overlap, and the exchange shrinks by that factor (C4: 715 MB ->
    a few tens of MB per rank)"""
    off = PAIR_RESULT.fields["accepted"][1]
    acc = local[:, off:off + 4].contiguous().view(torch.int32).reshape(-1) != 0
    return local[acc]


def allgather_pair_results(local, n_local_max=None, accepted_only=False):
    """local: uint8 tensor [n_local, 9664] (device for nccl, cpu for gloo).  Returns uint8 [world, n_max, 9664] and the
    per-rank counts; ranks may hold different numbers of pairs (counts are gathered first, payload padded)."""
    if accepted_only:
        local = compact_accepted(local)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local.unsqueeze(0), [local.shape[0]]
    if dist.get_backend() == "gloo" and local.is_cuda:
        # gloo has no device all_gather: stage through the host (CPU tests and single-GPU dry runs of bench.py only;
        # production uses backend "nccl" = RCCL, device to device over xGMI)
        g, counts = allgather_pair_results(local.cpu(), n_local_max, False)
        return g.to(local.device), counts
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1) if n_local_max is None else n_local_max
    padded = local
    if local.shape[0] != n_max:
        padded = torch.zeros((n_max, REC), dtype=torch.uint8, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * n_max, REC), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    return out.view(world, n_max, REC), counts


def gathered_to_records(gathered, counts):
    """uint8 [world, n_max, 9664] (+ counts) -> one structured numpy array of all valid records, rank-major"""
    g = gathered.cpu().numpy()
    parts = [g[r, :c].reshape(-1).view(PAIR_RESULT) for r, c in enumerate(counts)]
    return np.concatenate(parts) if parts else np.zeros(0, PAIR_RESULT)
