"""Candidate-batch data parallelism over the GPUs of one node.

Every candidate's μ, σ² and score depend only on (X, y, hyper-parameters, that candidate)
(src/surrogates/StandardGP.jl:361-379, ExpectedImprovement.jl:41-44), so the candidate batch is cut
into `world` contiguous shards, one process per GPU.  Each rank fits the same surrogate redundantly
(deterministic kernels → bit-identical L and α; 0.2 % of the per-GPU work at N = 8192) and scores
its shard with no data-path collective.  The only exchange is the selection: each rank contributes
its local top-k as k × (score, global index) and every rank merges the gathered lists with the
reference's order (descending score, ties → lowest index: stable `sortperm(...; rev=true)`,
acq_utils.jl:51).  RCCL has no MAXLOC, hence all_gather + local merge instead of all_reduce; the
payload is 16·k bytes per rank — latency-bound on xGMI, bandwidth irrelevant.
"""
from __future__ import annotations

import numpy as np


def shard_range(M: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of M candidates for `rank`: sizes differ by at most one, earlier
    ranks take the larger shards."""
    base, extra = divmod(M, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_topk(vals: np.ndarray, idx: np.ndarray, k: int):
    """Merge gathered (score, global index) pairs into the global top-k in Julia's stable
    reverse-sortperm order; entries with index −1 (padding of short shards) are dropped."""
    vals = np.asarray(vals, dtype=np.float64).reshape(-1)
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    keep = idx >= 0
    vals, idx = vals[keep], idx[keep]
    isn = np.isnan(vals)
    key = np.where(isn, np.inf, vals)
    order = np.lexsort((idx, -key, ~isn))      # NaN first, then descending score, then lowest index
    order = order[:k]
    return vals[order], idx[order]


def all_gather_topk(local_vals, local_idx, k: int, group=None):
    """all_gather of each rank's local top-k, then the identical merge on every rank.
    local_vals / local_idx: length-k tensors (CUDA → RCCL over xGMI; CPU → gloo) or NumPy arrays."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    was_numpy = not torch.is_tensor(local_vals)
    v = torch.as_tensor(local_vals, dtype=torch.float64)
    i = torch.as_tensor(local_idx, dtype=torch.int64)
    # one 16·k-byte message per rank: pack (score bits, index) into a single int64 buffer
    packed = torch.stack([v.view(torch.int64), i])
    if dist.get_backend(group) != "nccl":
        packed = packed.cpu()                    # gloo: host tensors
    out = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(out, packed, group=group)
    allp = torch.stack(out).cpu()
    vals = allp[:, 0, :].contiguous().view(torch.float64).numpy()
    idx = allp[:, 1, :].numpy()
    mv, mi = merge_topk(vals, idx, k)
    if was_numpy:
        return mv, mi
    return torch.from_numpy(mv), torch.from_numpy(mi)


def sharded_acquisition(acqf, surrogate, make_shard, M: int, k: int, group=None):
    """Score M candidates across all ranks and return the global top-k (values, global indices),
    identical on every rank.  `make_shard(lo, hi)` returns this rank's candidates [lo, hi) — a CUDA
    tensor keeps everything on the device."""
    import torch.distributed as dist

    from .acquisition import evaluate

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(M, rank, world)
    Z = make_shard(lo, hi)
    _, tv, ti = evaluate(acqf, surrogate, Z, k=k, idx_base=lo, return_scores=False)
    return all_gather_topk(tv, ti, k, group=group)
