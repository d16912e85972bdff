"""CPU, world_size 2 over gloo: the carrier sharding and the job-level reductions bench.py uses
for N > 1 (the data path itself has no collective)."""
import os
import sys

import numpy as np
import pytest

from tetraear_amd.shard import carrier_range


def test_carrier_range_partitions_exactly():
    for total in (1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = carrier_range(total, r, world)
                assert 0 <= lo <= hi <= total
                seen.extend(range(lo, hi))
            assert seen == list(range(total))
            sizes = [carrier_range(total, r, world)[1] - carrier_range(total, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    import bench
    from tetraear_amd.shard import carrier_range, reduce_job
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 5
    lo, hi = carrier_range(total, rank, world)
    # each rank demodulates ITS carriers (CPU oracle stands in for the device here) ...
    n_sym = 0
    digest = []
    for c in range(lo, hi):
        o = OracleSignalProcessor(2.4e6)
        out = o.process(synth.cu8_to_c128(synth.noise_cu8(6000, 100 + c)), 0)
        n_sym += len(out)
        digest.append((c, int(out.sum())))
    dist.barrier()
    # ... and only the bookkeeping is reduced
    t, s, bad = reduce_job(bench.TorchGroup(dist), 0.1 * (rank + 1), n_sym, n_failed=rank)
    assert bad == 1                    # rank 1's failed check is known on every rank
    q.put((rank, t, s, digest))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_job_matches_single_process():
    import torch.multiprocessing as mp
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = {}
    total = 0
    for c in range(5):
        o = OracleSignalProcessor(2.4e6)
        out = o.process(synth.cu8_to_c128(synth.noise_cu8(6000, 100 + c)), 0)
        expect[c] = int(out.sum())
        total += len(out)
    got = {}
    for rank, t, s, digest in res:
        assert abs(t - 0.2) < 1e-12        # max over ranks
        assert s == total                  # sum over ranks
        got.update(dict(digest))
    assert got == expect


def test_strong_scaling_partition_covers_the_batch_once():
    """bench.py --total-carriers: every rank generates ITS slice of the one job (carriers lo .. hi - 1, seeds 1000 + g); the
    slices are disjoint, in order, complete, and equal to the job generated in one piece.  Weak scaling: rank r's batch is
    the job's carriers r * C .. r * C + C - 1, so the ranks' streams never repeat."""
    import bench
    from tetraear_amd.shard import carrier_range
    total, chunk = 21, 512
    iq, foffs = bench.make_batch(total, chunk, "cu8", 0)
    assert len(np.unique(iq.reshape(total, -1), axis=0)) == total      # every carrier its own stream
    for world in (1, 2, 4, 8):
        parts, offs = [], []
        for r in range(world):
            lo, hi = carrier_range(total, r, world)
            p, o = bench.make_batch(hi - lo, chunk, "cu8", lo)
            parts.append(p)
            offs.append(o)
        assert np.array_equal(np.concatenate(parts), iq) and np.array_equal(np.concatenate(offs), foffs)
        sizes = [len(o) for o in offs]
        assert max(sizes) - min(sizes) <= 1
    a, _ = bench.make_batch(4, chunk, "cu8", 0)
    b, _ = bench.make_batch(4, chunk, "cu8", 4)
    assert np.array_equal(np.concatenate([a, b]), bench.make_batch(8, chunk, "cu8", 0)[0])
