#!/usr/bin/env python3
"""Measured HBM ceilings of the box through the library's own copy / read / write kernels (tdm_hbm_ceiling; no tensor
framework).  usage: tools/hbm_ceiling.py [out.json]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetraear_amd import _lib  # noqa: E402

L = _lib.load()
out = {}
for gib in (1, 2):
    g = (C.c_double * 3)()
    _lib.check(L.tdm_hbm_ceiling(0, gib << 30, 20, g))
    out[f"{gib}GiB"] = {"copy_GBps": g[0], "read_GBps": g[1], "write_GBps": g[2]}
res = {"what": "tdm_hbm_ceiling: 16 B per lane; flat form (one access per lane, one workgroup per 4 KB) and grid-stride forms (plain / "
               "non-temporal, 2048 / 8192 workgroups); best of each; 20 timed launches after 3; copy counts bytes read + bytes written",
       "buffers": out, "spec_GBps": 8000.0,
       "ceiling_GBps": max(v["copy_GBps"] for v in out.values())}
print(json.dumps(res))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
