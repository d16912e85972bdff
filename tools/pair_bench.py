#!/usr/bin/env python3
"""Experiment (GPU box): the headline batch as TWO (or four) plans of a part of the carriers each, every plan on its own
stream, against one plan: the small launches and the low-rate stage of one plan overlap the other plan's decimator.
Round 4 also tried to steer the overlap -- the plans' decimators made to alternate through events, the low-rate stage on
a high-priority stream, dynamic LDS to cap either kernel's occupancy so that one wavefront of each shares a SIMD
(256 + 240 of 512 VGPRs) -- none of which beat plain two streams.  usage: pair_bench.py [carriers]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from tetraear_amd import _lib  # noqa: E402
from tetraear_amd.batch import BatchDemodulator  # noqa: E402

carriers = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk, steps = 262144, 200
u8, foffs = bench.make_batch(carriers, chunk, "cu8", 0)
lib = _lib.load()


def run(parts, pair):
    per = carriers // parts
    bds = []
    for i in range(parts):
        bd = BatchDemodulator(bench.SAMPLE_RATE, chunk, per, "cu8")
        bd.alloc_device_io()
        bd.upload(u8[2 * chunk * per * i: 2 * chunk * per * (i + 1)], freq_offsets=foffs[per * i: per * (i + 1)])
        bds.append(bd)
    for _ in range(200):
        for bd in bds:
            bd.enqueue()
    for bd in bds:
        bd.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for bd in bds:
            bd.enqueue()
    for bd in bds:
        bd.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    outs = [bd.download() for bd in bds]
    hard = np.concatenate([o[0] for o in outs]); n_soft = np.concatenate([o[2] for o in outs]); bp = np.concatenate([o[3] for o in outs])
    digest = bench.output_digest(hard, n_soft, bp)
    for bd in bds:
        bd.close()
    return ms, digest


want = bench.expected_digest(bench.digest_key(carriers, chunk, "cu8", bench.SAMPLE_RATE, 0, False))
configs = (("one plan", 1, False), ("two plans, two streams", 2, False), ("four plans, four streams", 4, False))
if len(sys.argv) > 2:
    configs = tuple(c for c in configs if c[0].startswith(sys.argv[2]))
for name, parts, pair in configs:
    ms, d = run(parts, pair)
    nsym = carriers * 2015
    print(json.dumps({"config": name, "ms_per_step": ms, "Msym_s": nsym / ms / 1e3, "digest_ok": d == want}), flush=True)
