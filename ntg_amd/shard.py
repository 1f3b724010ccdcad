"""Sharding of a batch of independent problems over ranks, and the one collective of the path.

Problems are independent (no iteration-time exchange): rank r of W owns a contiguous slice of the global problem stream; the only
collective is the final gather of (C*, objective) -- RCCL all-gather over xGMI on the GPU box (`torch.distributed` backend "nccl"),
gloo in the CPU rehearsal (tests/test_multi_rank.py).  bench.py and the tests share these functions: the test exercises the code the
benchmark runs, not a restatement of it."""
from __future__ import annotations


def rank_slice(total: int, world: int, rank: int) -> slice:
    """Contiguous slice of `total` problems owned by `rank` (sizes differ by at most one; every problem has exactly one owner)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def per_rank_counts(total: int, world: int) -> list[int]:
    return [rank_slice(total, world, r).stop - rank_slice(total, world, r).start for r in range(world)]


def gather_results(x, objective, total: int, world: int):
    """All-gather the per-rank solutions x [n_r, nC] and objectives [n_r] into [total, nC] / [total] on every rank.
    Equal slices: one all_gather_into_tensor per array; ragged slices are padded to the largest slice first."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return x, objective
    counts = per_rank_counts(total, world)
    if len(set(counts)) == 1:
        gx = torch.empty((total, x.shape[1]), dtype=x.dtype, device=x.device)
        go = torch.empty(total, dtype=objective.dtype, device=objective.device)
        dist.all_gather_into_tensor(gx, x.contiguous())
        dist.all_gather_into_tensor(go, objective.contiguous())
        return gx, go
    # ragged slices (they differ by at most one problem): pad to the largest, gather, drop the padding rows
    cmax = max(counts)
    px = torch.zeros((cmax, x.shape[1]), dtype=x.dtype, device=x.device); px[:x.shape[0]] = x
    po = torch.zeros(cmax, dtype=objective.dtype, device=objective.device); po[:objective.shape[0]] = objective
    gx = torch.empty((world * cmax, x.shape[1]), dtype=x.dtype, device=x.device)
    go = torch.empty(world * cmax, dtype=objective.dtype, device=objective.device)
    dist.all_gather_into_tensor(gx, px)
    dist.all_gather_into_tensor(go, po)
    keep = torch.cat([torch.arange(r * cmax, r * cmax + c, device=x.device) for r, c in enumerate(counts)])
    return gx[keep], go[keep]


def max_over_ranks(seconds: float, world: int, device=None) -> float:
    """The slowest rank's time (the job's time)."""
    if world == 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def imbalance(work_per_problem, world: int) -> dict:
    """Load imbalance of contiguous slices for a per-problem work measure (e.g. major iterations of a to-convergence solve):
    max over ranks of the slice's total work divided by the mean -- what a fixed slicing costs against a perfect balance."""
    import numpy as np
    w = np.asarray(work_per_problem, dtype=np.float64)
    tot = [float(w[rank_slice(len(w), world, r)].sum()) for r in range(world)]
    mean = sum(tot) / world
    return {"world": world, "max_over_mean": max(tot) / mean if mean > 0 else 1.0, "per_rank_total": tot}
