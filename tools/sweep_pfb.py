"""Randomised channeliser sweep on the GPU: tdm_channelise(_batch) vs oracle/pfb_np.py (fp64 definition)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import pfb_np
from tetraear_amd import synth
from tetraear_amd.channeliser import channelise, channelise_batch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time(); bad = 0; cnt = 0
while time.time() - t0 < budget:
    M = int(rng.choice([72, 80, 96, 128, 400]))
    D = int(rng.integers(max(2, M // 8), M + 1)) if rng.random() < 0.6 else int(rng.choice([M // 4, M // 3, M // 2, 125 if M == 400 else M // 3]))
    D = max(D, 1)
    n = int(rng.integers(1, 9000))
    fmt = str(rng.choice(["cf32", "cu8", "cs8"]))
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.2
    if fmt == "cf32":
        raw = x.astype(np.complex64); xd = raw.astype(np.complex128)
    elif fmt == "cu8":
        raw = synth.quantise_cu8(x, scale=1.0); xd = synth.cu8_to_c128(raw)
    else:
        q = np.clip(np.round(np.stack([x.real, x.imag], -1) * 128), -128, 127).astype(np.int8); raw = q.reshape(-1)
        xd = (q[:, 0].astype(np.float64) + 1j * q[:, 1].astype(np.float64)) / 128.0
    pitch = 0 if rng.random() < 0.5 else ((n + D - 1) // D + 15) // 16 * 16
    y = channelise_batch(raw, fmt, 1, M, D, pitch=pitch)[0]
    probe = sorted(set(int(k) for k in rng.integers(0, M, 5)) | {0, M - 1})
    ref = pfb_np.channelise(xd, M, D, channels=probe)
    sc = max(np.max(np.abs(ref)), 1e-30)
    err = max(np.max(np.abs(y[k] - ref[i])) for i, k in enumerate(probe))
    cnt += 1
    if not err < 2e-5 * sc:
        bad += 1; print("MISMATCH", M, D, n, fmt, pitch, err / sc)
print(f"{cnt} cases, {bad} mismatches")
