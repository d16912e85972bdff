#!/usr/bin/env python3
"""Runs TDM_MODE_TETRA_GARDNER on the tetra bench workload a few times (for rocprofv3 / counter passes).
usage: run_gardner.py [rows] [steps] [sample rate: the bench rows are then read as if sampled at it -- timing only]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from tetraear_amd._lib import MODE_TETRA_GARDNER  # noqa: E402
from tetraear_amd.batch import BatchDemodulator  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
base = bench.tetra_rows()
fs = float(sys.argv[3]) if len(sys.argv) > 3 else bench.TETRA_FS
import os
if "GSEG" in os.environ:      # pieces per chunk: 0 whole chunks, 1 the plan's choice, K at most K (tdm_debug_set)
    from tetraear_amd import _lib
    _lib.load().tdm_debug_set(b"gardner_segments", int(os.environ["GSEG"]))
bd = BatchDemodulator(fs, bench.TETRA_N, rows, "cf32", mode=MODE_TETRA_GARDNER)
bd.alloc_device_io()
bd.upload(np.concatenate([base[i % 8] for i in range(rows)]))
for _ in range(steps):
    bd.enqueue()
bd.sync()
bd.time_begin()
for _ in range(steps):
    bd.enqueue()
print(bd.time_end() / steps, bd.stage_times())
out = bd.download()
import hashlib
print('n_soft', out[2][:8], 'sha', hashlib.sha256(out[0].tobytes()).hexdigest()[:12])
np.savez(sys.argv[4], hard=out[0][:64], soft=out[1][:64], n_soft=out[2][:64], bp=out[3][:64]) if len(sys.argv) > 4 else None
bd.close()
