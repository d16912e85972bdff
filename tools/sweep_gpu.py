"""Randomised length / rate sweep on the GPU: SignalProcessor.process_cu8 vs the C oracle; since round 6 one case in ten goes
through process() as complex128 / complex64 with a NaN or an Inf put somewhere into the chunk (the reference's all-NaN chunk
wherever a zero-phase filter runs: tests/test_nonfinite.py)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle.oracle import OracleSignalProcessor
from tetraear_amd import synth
from tetraear_amd.signal.processor import SignalProcessor
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time(); bad = 0; cnt = 0
rates = [2.4e6, 2.4e6, 2.4e6, 1.8e6, 2.048e6, 960000.0, 480000.0, 240000.0, 72000.0, 3.2e6, 10e6]
specials = [27, 28, 29, 150, 160, 161, 5119, 5120, 5121, 5271, 10240, 20473, 20480, 20481, 2048, 4096, 4097, 6143, 6144, 6145, 131072, 262144, 131071, 262145, 200000]
procs = {}
while time.time() - t0 < budget:
    fs = rates[rng.integers(len(rates))]
    n = int(specials[rng.integers(len(specials))]) if rng.random() < 0.4 else int(rng.integers(1, 300000))
    f = 0.0 if rng.random() < 0.3 else float(rng.uniform(-8000, 8000))
    u8 = synth.noise_cu8(n, int(rng.integers(1 << 30)))
    x = synth.cu8_to_c128(u8)
    ref = OracleSignalProcessor(fs)
    r = ref.process(x, f)
    p = procs.setdefault(fs, SignalProcessor(fs))
    if rng.random() < 0.1 and n > 1:
        xx = x.astype(np.complex64) if rng.random() < 0.3 else x.copy()
        for _ in range(int(rng.integers(1, 3))):
            v = [np.nan, np.inf, -np.inf, complex(0.5, np.nan), complex(np.inf, -1.0)][int(rng.integers(5))]
            xx[int(rng.integers(n))] = v
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = ref.process(xx.astype(np.complex128), f)
            h = p.process(xx, freq_offset=f)
        cnt += 1; nonfinite = nonfinite + 1 if "nonfinite" in dir() else 1
        a, b = np.asarray(p.symbols, dtype=complex), np.asarray(ref.symbols, dtype=complex)
        ok = len(h) == len(r) and np.array_equal(h, r) and len(a) == len(b) and np.array_equal(np.isnan(a.real), np.isnan(b.real)) and np.array_equal(np.isnan(a.imag), np.isnan(b.imag))
        if ok and len(b) and np.isfinite(b).any():
            fin = np.isfinite(b)
            ok = np.array_equal(np.isfinite(a), fin) and np.max(np.abs(a[fin] - b[fin])) <= (5e-5 if xx.dtype == np.complex64 else 1e-10) * max(np.max(np.abs(b[fin])), 1e-300)
        if not ok:
            bad += 1; print("MISMATCH (non-finite)", fs, n, f, xx.dtype, len(h), len(r))
        continue
    h = p.process_cu8(u8, freq_offset=f); cnt += 1
    ok = len(h) == len(r) and np.array_equal(h, r) and len(p.symbols) == len(ref.symbols)
    if ok and len(ref.symbols):
        sc = np.max(np.abs(ref.symbols)) or 1.0
        ok = np.max(np.abs(p.symbols - ref.symbols)) <= 1e-10 * sc
    if not ok:
        bad += 1; print("MISMATCH", fs, n, f, len(h), len(r))
print(f"{cnt} cases ({nonfinite if 'nonfinite' in dir() else 0} of them with non-finite samples), {bad} mismatches")
