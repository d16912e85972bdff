#!/usr/bin/env python3
"""Experiment (GPU box): BASELINE config 5 end to end with the batch worked off in sub-batches of G streams whose channel
rows stay in the Infinity Cache (WidebandReceiver(group=G)).  Prints ms per 32-stream batch for G and slot counts."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from tetraear_amd.wideband import WidebandReceiver  # noqa: E402

streams, steps = 32, 60
u8, _ = bench.wideband_stream()
want = str(bench.load_checks()["wideband_digest"])
occupied = [int(k) for k in bench.WIDEBAND_CHANNELS]
res = []
for G in (32, 16, 8, 4, 2):
    for slots in (1, 2):
        rx = WidebandReceiver(bench.PFB_FS, bench.PFB_NIN, bench.PFB_M, bench.PFB_D, streams=streams, fmt="cu8", slots=slots,
                              group=None if G == streams else G)
        rx.d_in.upload(np.tile(u8, streams))
        if G == streams:
            run = lambda k: rx.enqueue(slot=k % slots)
        else:
            run = lambda k: rx.enqueue_grouped()
        for k in range(160):
            run(k)
        rx.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            run(k)
        rx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        if G == streams:
            hard, soft, n_soft, bp, mm = rx.demod.download()
        else:
            hard, soft, n_soft, bp, mm = rx.download_grouped()
        ok = all(bench.rows_digest(hard, n_soft, [s * bench.PFB_M + k for k in occupied]) == want for s in range(streams))
        res.append({"group": G, "slots": slots, "ms_per_batch": ms, "digest_ok": bool(ok)})
        print(res[-1], flush=True)
        rx.close()
json.dump(res, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None
