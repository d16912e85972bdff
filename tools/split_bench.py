#!/usr/bin/env python3
"""Experiment (GPU box, round 6): how to keep the device busy behind the decimator -- more than one plan, each on its own
stream.  Candidates: plans of the WHOLE batch taking the steps in turn (two / three steps in flight: what
tetraear_amd.batch.PipelinedBatchDemodulator does); one plan; the batch cut into two halves of the carriers.  (The first
version also tried four quarters and "whole dispatch rounds on the raw-byte kernel + the rest on the double-based kernel":
both slower than one plan, profiles/r06_ab/split_spatial.txt.)
usage: split_bench.py C [steps]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from tetraear_amd import _lib  # noqa: E402
from tetraear_amd._lib import debug_option  # noqa: E402
from tetraear_amd.batch import BatchDemodulator  # noqa: E402

carriers = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
chunk = 262144
u8, foffs = bench.make_batch(carriers, chunk, "cu8", 0)
want = bench.expected_digest(bench.digest_key(carriers, chunk, "cu8", bench.SAMPLE_RATE, 0, False))


def run(parts):
    """parts: list of (count, no_raw)"""
    bds, lo = [], 0
    for cnt, no_raw in parts:
        with debug_option("no_raw", 1 if no_raw else 0):
            bd = BatchDemodulator(bench.SAMPLE_RATE, chunk, cnt, "cu8")
        bd.alloc_device_io()
        bd.upload(u8[2 * chunk * lo: 2 * chunk * (lo + cnt)], freq_offsets=foffs[lo: lo + cnt])
        bds.append(bd)
        lo += cnt
    for _ in range(200):
        for bd in bds:
            bd.enqueue()
    for bd in bds:
        bd.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            for bd in bds:
                bd.enqueue()
        for bd in bds:
            bd.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    outs = [bd.download() for bd in bds]
    eng = [int(bd.info.dec_engine) for bd in bds]
    for bd in bds:
        bd.close()
    hard = np.concatenate([o[0] for o in outs]); n_soft = np.concatenate([o[2] for o in outs]); bp = np.concatenate([o[3] for o in outs])
    return best, bench.output_digest(hard, n_soft, bp) == want if want else None, eng


def run_alternate(depth):
    """`depth` plans of the WHOLE batch each, steps handed to them in turn (two steps in flight on two streams)"""
    bds = []
    for _ in range(depth):
        bd = BatchDemodulator(bench.SAMPLE_RATE, chunk, carriers, "cu8")
        bd.alloc_device_io()
        bd.upload(u8, freq_offsets=foffs)
        bds.append(bd)
    for k in range(200):
        bds[k % depth].enqueue()
    for bd in bds:
        bd.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps):
            bds[k % depth].enqueue()
        for bd in bds:
            bd.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    oks = []
    for bd in bds:
        hard, soft, n_soft, bp, mm = bd.download()
        oks.append(bench.output_digest(hard, n_soft, bp) == want if want else None)
        bd.close()
    return best, all(oks) if want else None, [3] * depth


slots, per = 2048, 35
configs = [("one plan", [(carriers, False)])]
if carriers >= 2:
    configs.append(("two halves", [(carriers // 2, False), (carriers - carriers // 2, False)]))
if carriers >= 4:
    q = carriers // 4
    configs.append(("four quarters", [(q, False)] * 3 + [(carriers - 3 * q, False)]))
rounds = carriers * per // slots
if rounds >= 1:
    full = rounds * slots // per
    if 0 < full < carriers:
        configs.append((f"{rounds} whole rounds ({full}) + rest ({carriers - full}) on doubles", [(full, False), (carriers - full, True)]))
        configs.append((f"{rounds} whole rounds ({full}) + rest ({carriers - full}) raw", [(full, False), (carriers - full, False)]))
        h = full // 2
        configs.append((f"whole rounds in two halves ({h}+{full - h}) + rest on doubles", [(h, False), (full - h, False), (carriers - full, True)]))
for depth in (2, 3):
    ms, ok, eng = run_alternate(depth)
    print(json.dumps({"carriers": carriers, "config": f"{depth} plans of the whole batch, steps in turn", "ms_per_step": round(ms, 5), "Msym_s": round(carriers * 2015 / ms / 1e3, 1), "digest_ok": ok}), flush=True)
for name, parts in configs[:2]:
    ms, ok, eng = run(parts)
    print(json.dumps({"carriers": carriers, "config": name, "ms_per_step": round(ms, 5), "Msym_s": round(carriers * 2015 / ms / 1e3, 1), "digest_ok": ok, "engines": eng}), flush=True)
