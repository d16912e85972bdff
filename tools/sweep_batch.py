"""Randomised batched sweep on the GPU: BatchDemodulator (all wire formats, independent and shared-stream
carriers with input-rate pre-shifts and AFC offsets) against the C oracle's composition
process(frequency_shift(x, pre), freq_offset)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle.oracle import OracleSignalProcessor
from tetraear_amd import synth
from tetraear_amd.batch import BatchDemodulator
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time(); bad = 0; cnt = 0
while time.time() - t0 < budget:
    fs = float(rng.choice([2.4e6, 2.4e6, 1.8e6, 2.048e6, 960000.0, 240000.0, 10e6]))
    n = int(rng.integers(30, 60000))
    rows = int(rng.integers(1, 6))
    fmt = str(rng.choice(["cu8", "cs8", "cf32", "cf64"]))
    shared = bool(rng.random() < 0.4)
    nstreams = 1 if shared else rows
    x = (rng.standard_normal((nstreams, n)) + 1j * rng.standard_normal((nstreams, n))) * 0.3
    if fmt == "cu8":
        raw = np.stack([synth.quantise_cu8(x[i], scale=1.0) for i in range(nstreams)]); xd = np.stack([synth.cu8_to_c128(raw[i]) for i in range(nstreams)])
    elif fmt == "cs8":
        q = np.clip(np.round(np.stack([x.real, x.imag], -1) * 128), -128, 127).astype(np.int8); raw = q.reshape(nstreams, -1)
        xd = (q[..., 0].astype(np.float64) + 1j * q[..., 1].astype(np.float64)) / 128.0
    elif fmt == "cf32":
        raw = x.astype(np.complex64); xd = raw.astype(np.complex128)
    else:
        raw = x.astype(np.complex128); xd = raw
    pre = rng.uniform(-3e5, 3e5, rows) if (shared or rng.random() < 0.3) else None
    fo = rng.uniform(-8000, 8000, rows) if rng.random() < 0.7 else None
    bd = BatchDemodulator(fs, n, rows, fmt)
    hards, softs, bp, mm = bd.process(raw, freq_offsets=fo, pre_shifts=pre, shared_input=shared)
    for r in range(rows):
        ref = OracleSignalProcessor(fs)
        xi = xd[0 if shared else r]
        if pre is not None:
            xi = ref.frequency_shift(xi, float(pre[r]))
        h = ref.process(xi, 0.0 if fo is None else float(fo[r]))
        cnt += 1
        ok = np.array_equal(hards[r], h) and len(softs[r]) == len(ref.symbols)
        if ok and len(ref.symbols):
            ok = np.max(np.abs(softs[r] - ref.symbols)) <= 1e-9 * (np.max(np.abs(ref.symbols)) or 1.0)
        if not ok:
            bad += 1; print("MISMATCH", fs, n, rows, fmt, shared, r)
    bd.close()
print(f"{cnt} carriers, {bad} mismatches")
