#!/usr/bin/env python3
"""Anchors the FIR arithmetic of the two TETRA-mode definitions to scipy.signal, the reference's own named dependency
(requirements.txt: scipy>=1.10; the reference itself has no channeliser and no RRC filter):

  channeliser   channel k of the polyphase bank = mix by exp(-2 pi i k n / M) -> prototype FIR -> keep every D-th sample,
                computed per channel with scipy.signal.upfirdn; the prototype itself with scipy.signal.firwin
                (Kaiser window, beta 8, cutoff at half the output rate, unit DC gain)
  matched filter  scipy.signal.lfilter(h, 1, x) with the group delay taken out (the RRC taps are the definition's own:
                scipy has no root-raised-cosine design), then the definition's remaining steps on THAT output

    python tests/golden/make_golden_scipy_fir.py        (scipy 1.15.3 / numpy 2.2.6 here; versions stored)

Writes tests/golden/scipy_fir.npz: inputs as bytes (cu8 / complex64) and scipy's outputs.  tests/test_scipy_anchor.py
holds the numpy definitions (CPU tier) and the device (GPU tier) to these.
"""
import os
import sys

import numpy as np
import scipy
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tetra_np  # noqa: E402
from tetraear_amd import synth  # noqa: E402

PFB_CASES = [(400, 125, 20000, 11), (96, 32, 12000, 12), (128, 40, 9000, 13)]   # M, D, n_in, seed
RRC_CASES = [(72000.0, 12000, 21), (80000.0, 9000, 22), (54000.0, 8000, 23)]     # fs, n, seed


def main():
    out = {"scipy_version": np.array(scipy.__version__), "numpy_version": np.array(np.__version__)}
    for ci, (M, D, n, seed) in enumerate(PFB_CASES):
        u8 = synth.noise_cu8(n, seed)
        x = synth.cu8_to_c128(u8)
        h = signal.firwin(3 * M, 1.0 / D, window=("kaiser", 8.0))          # 3 taps per branch
        probe = np.array(sorted({0, 1, M // 3, M // 2, M - 2, M - 1}), dtype=np.int64)
        n_out = (n + D - 1) // D
        ys = []
        for k in probe:
            xs = x * np.exp(-2j * np.pi * ((int(k) * np.arange(n)) % M) / M)
            ys.append(signal.upfirdn(h, xs, up=1, down=D)[:n_out])
        out[f"pfb{ci}_u8"] = u8
        out[f"pfb{ci}_MD"] = np.array([M, D], dtype=np.int64)
        out[f"pfb{ci}_h"] = h
        out[f"pfb{ci}_probe"] = probe
        out[f"pfb{ci}_y"] = np.stack(ys)
    for ci, (fs, n, seed) in enumerate(RRC_CASES):
        x, _ = synth.dqpsk_baseband(n, fs, seed, timing_offset=0.13 * ci)
        rng = np.random.default_rng(seed + 100)
        x = (x + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        sps = fs / 18000.0
        h = tetra_np.rrc_taps(sps, exact=True)                             # the unquantised taps
        half = len(h) // 2
        y = signal.lfilter(h, [1.0], np.concatenate([x.astype(np.complex128), np.zeros(half)]))[half:]
        # the rest of the definition on scipy's filter output: timing estimates, symbol instants, Farrow
        tau_b = tetra_np.timing_estimates(y, sps)
        kmax = int(np.floor(n / sps)) + 2
        k = np.arange(0, kmax, dtype=np.float64)
        t = (k + tetra_np.tau_of_sample(k * sps, tau_b)) * sps
        t = t[(t >= 1.0) & (t <= n - 3.0)]
        out[f"rrc{ci}_x"] = x
        out[f"rrc{ci}_fs"] = np.float64(fs)
        out[f"rrc{ci}_y"] = y
        out[f"rrc{ci}_sym"] = tetra_np.farrow(y, t)
    np.savez_compressed(os.path.join(HERE, "scipy_fir.npz"), **out)
    print("wrote scipy_fir.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
