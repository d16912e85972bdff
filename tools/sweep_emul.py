"""Randomised length / rate sweep: kernel bodies in CPU emulation vs the C oracle (block-boundary coverage)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.emul import emul
from oracle.oracle import OracleSignalProcessor
from tetraear_amd import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); bad = 0; cnt = 0
rates = [2.4e6, 2.4e6, 2.4e6, 1.8e6, 2.048e6, 960000.0, 480000.0, 240000.0, 72000.0, 3.2e6]
specials = [27, 28, 29, 150, 160, 161, 5120 - 1, 5120, 5120 + 1, 5121 + 150, 10240, 20480 - 7, 20480, 20481, 2048, 4096, 4097, 6143, 6144, 6145]
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    fs = rates[rng.integers(len(rates))]
    n = int(specials[rng.integers(len(specials))]) if rng.random() < 0.4 else int(rng.integers(1, 30000))
    f = 0.0 if rng.random() < 0.3 else float(rng.uniform(-8000, 8000))
    u8 = synth.noise_cu8(n, int(rng.integers(1 << 30)))
    x = synth.cu8_to_c128(u8)
    ref = OracleSignalProcessor(fs)
    r = ref.process(x, f)
    hard, soft, n_soft, bp, mm = emul.process(fs, u8, "cu8", n, 1, freq_offset=np.array([f]))
    ns = int(n_soft[0]); cnt += 1
    ok = ns == len(ref.symbols) and np.array_equal(hard[0, :max(ns - 1, 0)], r)
    if ok and ns:
        sc = np.max(np.abs(ref.symbols)) or 1.0
        ok = np.max(np.abs(soft[0, :ns] - ref.symbols)) <= 1e-10 * sc
    if not ok:
        bad += 1; print("MISMATCH", fs, n, f, ns, len(ref.symbols))
print(f"{cnt} cases, {bad} mismatches")
