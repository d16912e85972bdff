#!/usr/bin/env python3
"""Runs on the GPU box.  For the default bench workloads (bench.py make_batch, ranks 0..7) run the batch once through
BatchDemodulator.enqueue, compare EVERY carrier with the CPU oracle (hard symbols and timing phase equal, soft
<= 1e-10) and only then write the digest bench.py asserts (tests/golden/bench_digest.json).
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


def check_batch(carriers, chunk, rank, rate=bench.SAMPLE_RATE):
    u8, foffs = bench.make_batch(carriers, chunk, "cu8", rank)
    bd = BatchDemodulator(rate, chunk, carriers, "cu8")
    bd.alloc_device_io()
    bd.upload(u8, freq_offsets=foffs)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    cache, worst = {}, 0.0
    for r in range(carriers):
        key = (r % min(carriers, 8), float(foffs[r]))
        if key not in cache:
            o = OracleSignalProcessor(rate)
            ref = o.process(synth.cu8_to_c128(u8[2 * chunk * r: 2 * chunk * (r + 1)]), foffs[r])
            cache[key] = (ref, o.symbols.copy(), o.best_phase)
        ref, sym, phase = cache[key]
        ns = int(n_soft[r])
        assert ns == len(sym) and bp[r] == phase, (rank, r)
        assert np.array_equal(hard[r, :ns - 1], ref), (rank, r)
        worst = max(worst, float(np.max(np.abs(soft[r, :ns] - sym)) / np.max(np.abs(sym))))
    assert worst <= 1e-10, worst
    return bench.output_digest(hard, n_soft, bp), len(cache), worst


def check_shared(carriers, chunk, rate=bench.SAMPLE_RATE):
    """bench.py --shared: ONE stream, every carrier shifted out of it; every carrier against
    p.process(p.frequency_shift(x, f_k)) of the oracle"""
    iq, _ = bench.make_batch(1, chunk, "cu8", 0)
    pre = bench.shared_offsets(carriers)
    bd = BatchDemodulator(rate, chunk, carriers, "cu8")
    bd.alloc_device_io(shared_input=True)
    bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    x = synth.cu8_to_c128(iq)
    worst = 0.0
    for r in range(carriers):
        o = OracleSignalProcessor(rate)
        ref = o.process(o.frequency_shift(x, pre[r]))
        ns = int(n_soft[r])
        assert ns == len(o.symbols) and bp[r] == o.best_phase, r
        assert np.array_equal(hard[r, :ns - 1], ref), r
        worst = max(worst, float(np.max(np.abs(soft[r, :ns] - o.symbols)) / np.max(np.abs(o.symbols))))
    assert worst <= 1e-10, worst
    return bench.output_digest(hard, n_soft, bp), carriers, worst


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
        for rank in range(ranks):
            d, n_or, worst = check_batch(1024, 262144, rank)
            res[bench.digest_key(1024, 262144, "cu8", bench.SAMPLE_RATE, rank, False)] = d
            print(f"rank {rank}: 1024 carriers equal to {n_or} oracle runs, soft err {worst:.2e}, sha256 {d[:16]}", flush=True)
        d, n_or, worst = check_batch(128, 262144, 0)
        res[bench.digest_key(128, 262144, "cu8", bench.SAMPLE_RATE, 0, False)] = d
    d, n_or, worst = check_shared(64, 262144)
    res[bench.digest_key(64, 262144, "cu8", bench.SAMPLE_RATE, 0, True)] = d
    print(f"shared: 64 carriers equal to {n_or} oracle runs, soft err {worst:.2e}, sha256 {d[:16]}", flush=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
