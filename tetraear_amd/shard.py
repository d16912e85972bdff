"""Work split of independent carriers across ranks (one process per GPU).

The path has no exchange step: carriers are partitioned statically and the
only cross-rank traffic is a barrier plus two tiny reductions (max elapsed time, sum of symbols).
`torch.distributed` is imported by the caller (bench.py) only when world_size > 1; backend "nccl"
is RCCL on ROCm, "gloo" is used by the CPU tests.
"""


def carrier_range(n_total, rank, world):
    """Contiguous block partition: rank r gets carriers [lo, hi); sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def reduce_job(dist, elapsed_s, n_symbols, device=None):
    """(max elapsed over ranks, total symbols over ranks). `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(n_symbols)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([int(n_symbols)], dtype=torch.int64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), int(s.item())
