"""Work split of independent carriers across ranks (one process per GPU).

The path has no exchange step: carriers are partitioned statically and the
only cross-rank traffic is a barrier plus three tiny reductions (max elapsed time, sum of symbols, sum of
output-check failures).  `group` is any object with max_f64 / sum_i64 / barrier:
tetraear_amd.rccl.RcclGroup (librccl through ctypes, what bench.py uses on GPUs) or a test double.
Nothing in this package imports a tensor framework.
"""


def carrier_range(n_total, rank, world):
    """Contiguous block partition: rank r gets carriers [lo, hi); sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def reduce_job(group, elapsed_s, n_symbols, n_failed=0):
    """(max elapsed over ranks, total symbols over ranks, total failed output checks over ranks).
    `group`: see the module docstring; None = single process.  Every rank must call it (the reductions are
    collective), in particular a rank whose own output check failed: the failure count travels with the job and every
    rank learns of it, so that all of them can stop together instead of one leaving the others inside an all-reduce."""
    if group is None:
        return float(elapsed_s), int(n_symbols), int(n_failed)
    return group.max_f64(elapsed_s), group.sum_i64(n_symbols), group.sum_i64(n_failed)
