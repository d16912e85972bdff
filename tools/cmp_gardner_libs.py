#!/usr/bin/env python3
"""temporary: outputs of TDM_MODE_TETRA_GARDNER under two builds of the library (subprocess per build), bit for bit"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    import bench
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    out = {}
    for rows, n in ((70, bench.TETRA_N), (5, 4999), (17, 16903)):
        base = bench.tetra_rows()
        bd = BatchDemodulator(bench.TETRA_FS, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        hards, softs, tm, mm = bd.process(np.concatenate([base[i % 8][:n] for i in range(rows)]))
        out[f"h{rows}"], out[f"s{rows}"], out[f"t{rows}"], out[f"m{rows}"] = np.concatenate(hards), np.concatenate(softs), tm, mm
        out[f"n{rows}"] = np.array([len(x) for x in softs])
        bd.close()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
paths = {"old": "tools/harness/libtetrahip_g1.so", "new": "tetraear_amd/libtetrahip.so"}
res = {}
for k, p in paths.items():
    f = f"/tmp/cmp_{k}.npz"
    subprocess.check_call([sys.executable, __file__, "child", f], env=dict(os.environ, TETRAHIP_LIB=os.path.abspath(p)))
    res[k] = np.load(f)
for key in res["old"].files:
    a, b = res["old"][key], res["new"][key]
    same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" or a.dtype.kind == "c" else np.array_equal(a, b)
    print(key, a.shape, "identical" if same else f"DIFFERS ({int(np.sum(a != b))} elements)")
