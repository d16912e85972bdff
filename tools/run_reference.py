#!/usr/bin/env python3
"""Runs the reference-mode path on the bench workload's shape a few times and prints the per-stage times (for A/B builds
and counter passes; no output check).  usage: run_reference.py [carriers] [steps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tetraear_amd.batch import BatchDemodulator  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bd = BatchDemodulator(2.4e6, 262144, rows, "cu8")
bd.alloc_device_io()
rng = np.random.default_rng(1)
bd.upload(rng.integers(0, 256, size=rows * 262144 * 2, dtype=np.uint8))
for _ in range(150):
    bd.enqueue()
bd.sync()
bd.time_begin()
for _ in range(steps):
    bd.enqueue()
print(bd.time_end() / steps, bd.stage_times())
bd.close()
