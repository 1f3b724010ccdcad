"""Sharding of a batch of independent problems over ranks, and the one collective of the path.

Problems are independent (no iteration-time exchange): rank r of W owns a contiguous slice of the global problem stream; the only
collective is the final gather of (C*, objective, inform, iters) -- RCCL all-gather over xGMI on the GPU box (`torch.distributed` backend "nccl"),
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


def gather_results(x, objective, total: int, world: int, inform=None, iters=None):
    """All-gather the per-rank results onto every rank: x [n_r, nC] and objective [n_r] -> [total, nC] / [total]; with `inform` and
    `iters` (int32 [n_r], SURVEY 8e lists all four) a 4-tuple (x, objective, inform, iters).
    ONE collective: the arrays travel as one packed fp64 record per problem, [C* | objective | inform | iters] (small integers are exact
    in fp64) -- a single all_gather_into_tensor over xGMI instead of one per array.  Ragged slices (they differ by at most one problem)
    are padded to the largest slice first and the padding rows dropped afterwards."""
    import torch
    import torch.distributed as dist
    full = inform is not None and iters is not None
    if world == 1:
        return (x, objective, inform, iters) if full else (x, objective)
    counts = per_rank_counts(total, world)
    cmax, nC = max(counts), x.shape[1]
    rec = torch.zeros((cmax, nC + 3), dtype=torch.float64, device=x.device)
    n = x.shape[0]
    rec[:n, :nC] = x
    rec[:n, nC] = objective
    if full:
        rec[:n, nC + 1] = inform.to(torch.float64)
        rec[:n, nC + 2] = iters.to(torch.float64)
    g = torch.empty((world * cmax, nC + 3), dtype=torch.float64, device=x.device)
    dist.all_gather_into_tensor(g, rec)
    if len(set(counts)) != 1:
        keep = torch.cat([torch.arange(r * cmax, r * cmax + c, device=x.device) for r, c in enumerate(counts)])
        g = g[keep]
    gx, go = g[:, :nC].to(x.dtype), g[:, nC].to(objective.dtype)
    if not full:
        return gx, go
    return gx, go, g[:, nC + 1].to(inform.dtype), g[:, nC + 2].to(iters.dtype)


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
