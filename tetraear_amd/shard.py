"""Work split of independent carriers across ranks (one process per GPU).

The path has no exchange step: carriers are partitioned statically and the
only cross-rank traffic is a barrier plus two tiny reductions (max elapsed time, sum of symbols).
`group` is any object with max_f64 / sum_i64 / barrier: tetraear_amd.rccl.RcclGroup (librccl through ctypes, what
bench.py uses on GPUs), or TorchGroup below around torch.distributed ("gloo" in the CPU tests, "nccl" = RCCL as the
fallback when the ctypes binding cannot initialise).
"""


def carrier_range(n_total, rank, world):
    """Contiguous block partition: rank r gets carriers [lo, hi); sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class TorchGroup:
    """the same three operations on an initialised torch.distributed process group"""

    def __init__(self, dist, device=None):
        self.dist, self.device = dist, device

    def max_f64(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_i64(self, x):
        import torch
        t = torch.tensor([int(x)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def barrier(self):
        self.dist.barrier()
        if self.device is not None:
            import torch
            torch.cuda.synchronize()

    def close(self):
        self.dist.destroy_process_group()


def reduce_job(group, elapsed_s, n_symbols, device=None):
    """(max elapsed over ranks, total symbols over ranks).  `group`: see the module docstring; None = single process;
    a torch.distributed module is accepted too (wrapped)."""
    if group is None:
        return float(elapsed_s), int(n_symbols)
    if hasattr(group, "all_reduce"):          # torch.distributed itself
        if not group.is_initialized() or group.get_world_size() == 1:
            return float(elapsed_s), int(n_symbols)
        group = TorchGroup(group, device)
    return group.max_f64(elapsed_s), group.sum_i64(n_symbols)
