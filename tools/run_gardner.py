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
bd.close()
