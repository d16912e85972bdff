#!/usr/bin/env python3
"""Runs on the GPU box.  For the default bench workloads (bench.py make_batch, ranks 0..7: 1024 DISTINCT streams each,
seeds 1000 + g) run the batch once through BatchDemodulator.enqueue, compare EVERY carrier with the CPU oracle (one
oracle run per carrier: hard symbols and timing phase equal, soft <= 1e-10) and only then write the digests bench.py
asserts (tests/golden/bench_digest.json), including the strong-scaling slices of rank 0's batch for 2..8 ranks.
usage: python tools/make_bench_digest.py <out.json> [ranks]"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench  # noqa: E402
from oracle.oracle import OracleSignalProcessor  # noqa: E402
from tetraear_amd import synth  # noqa: E402
from tetraear_amd.batch import BatchDemodulator  # noqa: E402


def check_batch(carriers, chunk, rank, rate=bench.SAMPLE_RATE, want_rows=False):
    """rank `rank`'s weak-scaling batch (the job's carriers rank * carriers ..., every one its own stream): ONE oracle run
    per carrier.  -> (digest, oracle runs, worst soft error[, (hard, n_soft, best_phase)])"""
    from concurrent.futures import ThreadPoolExecutor
    u8, foffs = bench.make_batch(carriers, chunk, "cu8", rank * carriers)
    bd = BatchDemodulator(rate, chunk, carriers, "cu8")
    bd.alloc_device_io()
    bd.upload(u8, freq_offsets=foffs)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()

    def one(r):   # (the C oracle runs outside the interpreter lock)
        o = OracleSignalProcessor(rate)
        ref = o.process(synth.cu8_to_c128(u8[2 * chunk * r: 2 * chunk * (r + 1)]), foffs[r])
        ns = int(n_soft[r])
        assert ns == len(o.symbols) and bp[r] == o.best_phase, (rank, r)
        assert np.array_equal(hard[r, :ns - 1], ref), (rank, r)
        return float(np.max(np.abs(soft[r, :ns] - o.symbols)) / np.max(np.abs(o.symbols)))
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as pool:
        errs = list(pool.map(one, range(carriers)))
    worst = max(errs)
    assert worst <= 1e-10, worst
    res = (bench.output_digest(hard, n_soft, bp), len(errs), worst)
    return res + ((hard, n_soft, bp),) if want_rows else res


def check_shared(carriers, chunk, rate=bench.SAMPLE_RATE, tchunks=1):
    """bench.py --shared [--chunks T]: ONE stream, every carrier shifted out of (every one of T consecutive chunks of) it;
    every plan row against p.process(p.frequency_shift(x_chunk, f_k)) of the oracle"""
    iq, _ = bench.make_shared_stream(chunk, tchunks, "cu8", 0)
    pre = np.tile(bench.shared_offsets(carriers), tchunks)
    bd = BatchDemodulator(rate, chunk, carriers * tchunks, "cu8")
    if tchunks > 1:
        bd.set_rows_per_chunk(carriers)
    bd.alloc_device_io(shared_input=tchunks == 1)
    bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    xs = synth.cu8_to_c128(iq)
    worst = 0.0
    for r in range(carriers * tchunks):
        o = OracleSignalProcessor(rate)
        x = xs[(r // carriers) * chunk: (r // carriers + 1) * chunk]
        ref = o.process(o.frequency_shift(x, pre[r]))
        ns = int(n_soft[r])
        assert ns == len(o.symbols) and bp[r] == o.best_phase, r
        assert np.array_equal(hard[r, :ns - 1], ref), r
        worst = max(worst, float(np.max(np.abs(soft[r, :ns] - o.symbols)) / np.max(np.abs(o.symbols))))
    assert worst <= 1e-10, worst
    return bench.output_digest(hard, n_soft, bp), carriers * tchunks, worst


if __name__ == "__main__":
    # usage: make_bench_digest.py <out.json> [ranks] [--shared-only]    (an existing out.json is updated, not replaced)
    out = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("--")]
    pos = [a for a in sys.argv[2:] if not a.startswith("--")]
    ranks = int(pos[0]) if pos else 8
    res = {}
    if os.path.exists(out):
        with open(out) as f:
            res = json.load(f)
    if "--shared-only" not in flags:
        from tetraear_amd.shard import carrier_range
        for rank in range(ranks):
            d, n_or, worst, rows = check_batch(1024, 262144, rank, want_rows=True)
            res[bench.digest_key(1024, 262144, "cu8", bench.SAMPLE_RATE, rank, False)] = d
            print(f"rank {rank}: 1024 carriers equal to {n_or} oracle runs, soft err {worst:.2e}, sha256 {d[:16]}", flush=True)
            if rank == 0:
                # BASELINE config 4's strong-scaling split of this very batch (bench.py --gpus N --total-carriers 1024): the
                # slices of the checked output, for every world size up to 8
                hard, n_soft, bp = rows
                for world in range(2, 9):
                    for r in range(world):
                        lo, hi = carrier_range(1024, r, world)
                        key = bench.digest_key(hi - lo, 262144, "cu8", bench.SAMPLE_RATE, r, False) + f":strong{lo}-{hi}of1024"
                        res[key] = bench.output_digest(hard[lo:hi], n_soft[lo:hi], bp[lo:hi])
        d, n_or, worst = check_batch(128, 262144, 0)
        res[bench.digest_key(128, 262144, "cu8", bench.SAMPLE_RATE, 0, False)] = d
    d, n_or, worst = check_shared(64, 262144)
    res[bench.digest_key(64, 262144, "cu8", bench.SAMPLE_RATE, 0, True)] = d
    print(f"shared: 64 carriers equal to {n_or} oracle runs, soft err {worst:.2e}, sha256 {d[:16]}", flush=True)
    d, n_or, worst = check_shared(64, 262144, tchunks=4)
    res[bench.digest_key(64, 262144, "cu8", bench.SAMPLE_RATE, 0, True) + ":chunks4"] = d
    print(f"shared, 4 consecutive chunks: 64 x 4 rows equal to {n_or} oracle runs, soft err {worst:.2e}, sha256 {d[:16]}", flush=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
