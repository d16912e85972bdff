#!/usr/bin/env python3
"""A/B of library configurations (tdm_debug_set switches) on the bench workload's shape: every configuration makes its own
plan, runs the same batch, and must give the same outputs bit for bit; times per step and per stage are printed.
usage: ab_switches.py [carriers] [steps] [chunk] -- name:key=val,key=val ...   (default: fused_carry 1 vs 0)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tetraear_amd import _lib, synth  # noqa: E402
from tetraear_amd.batch import BatchDemodulator  # noqa: E402

args = sys.argv[1:]
cfgs = []
if "--" in args:
    k = args.index("--")
    for c in args[k + 1:]:
        name, _, kv = c.partition(":")
        cfgs.append((name, dict((a.split("=")[0], int(a.split("=")[1])) for a in kv.split(",") if a)))
    args = args[:k]
if not cfgs:
    cfgs = [("default", {})]
rows = int(args[0]) if len(args) > 0 else 1024
steps = int(args[1]) if len(args) > 1 else 40
chunk = int(args[2]) if len(args) > 2 else 262144
nd = min(rows, 16)
base = np.stack([synth.dqpsk_cu8(chunk, 2.4e6, seed=1000 + i)[0] for i in range(nd)])
iq = np.concatenate([base[i % nd] for i in range(rows)])
import os
foff = ((np.arange(rows) % 7) - 3) * 390.625 * (0.0 if os.environ.get('ZERO_FOFF') else 1.0)
lib = _lib.load()
ref = None
ok = True
for rep in range(2):
    for name, sw in cfgs:
        old = {}
        for k, v in sw.items():
            import ctypes as C
            o = C.c_int64()
            _lib.check(lib.tdm_debug_get(k.encode(), C.byref(o)))
            old[k] = o.value
            _lib.check(lib.tdm_debug_set(k.encode(), v))
        bd = BatchDemodulator(2.4e6, chunk, rows, "cu8")
        bd.alloc_device_io()
        bd.upload(iq, freq_offsets=foff)
        for _ in range(80):
            bd.enqueue()
        bd.sync()
        bd.time_begin(per_stage=False)
        for _ in range(steps):
            bd.enqueue()
        total = bd.time_end() / steps
        bd.time_begin()
        for _ in range(steps):
            bd.enqueue()
        bd.time_end()
        st = bd.stage_times()
        out = bd.download()
        bd.close()
        for k, v in old.items():
            lib.tdm_debug_set(k.encode(), v)
        if ref is None:
            ref = out
        same = all(np.array_equal(x, y) for x, y in zip(ref, out))
        ok = ok and same
        print(f"{name:12s} {total:.4f} ms/step  {({k: round(v, 4) for k, v in st.items()})}  same_as_first={same}", flush=True)
import hashlib
print("symbols:", int(ref[2].sum()), "all identical:", ok, "sha256(hard,n_soft,bp):", hashlib.sha256(ref[0].tobytes() + ref[2].tobytes() + ref[3].tobytes()).hexdigest()[:16])
sys.exit(0 if ok else 1)
