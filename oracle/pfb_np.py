"""ORACLE for the tetra-mode channeliser (test infrastructure): numpy definition.

No reference implementation exists (the reference has no channeliser), so this is the
project's own definition -- "parity unpinned" for this stage; the HIP kernel is checked against it.

Uniform DFT filter bank, M channels spaced fs/M apart, every D-th output kept (oversampled when
D < M):      y_k[m] = sum_l h[l] x[mD - l] exp(-2 pi i k (mD - l) / M),   x[n] = 0 for n < 0,
i.e. channel k is frequency_shift(x, k fs/M) low-pass filtered by h and decimated by D.
Prototype h: Kaiser-windowed sinc, cutoff at half the output rate, M*P taps, unit DC gain.
"""
import numpy as np

PFB_TAPS_PER_BRANCH = 3
KAISER_BETA = 8.0


def prototype(M, D, P=PFB_TAPS_PER_BRANCH, beta=KAISER_BETA):
    L = M * P
    n = np.arange(L, dtype=np.float64) - (L - 1) / 2.0
    fc = 0.5 / D                      # cycles per input sample
    h = 2 * fc * np.sinc(2 * fc * n) * np.kaiser(L, beta)
    return h / np.sum(h)


def channelise(x, M, D, h=None, channels=None):
    """Direct evaluation (slow, for tests). Returns y[len(channels)][n_out], n_out = ceil(N/D)."""
    x = np.asarray(x, dtype=np.complex128)
    if h is None:
        h = prototype(M, D)
    N = len(x)
    n_out = (N + D - 1) // D
    if channels is None:
        channels = range(M)
    L = len(h)
    xp = np.concatenate([np.zeros(L - 1, dtype=np.complex128), x])
    n = np.arange(N, dtype=np.float64)
    out = np.zeros((len(list(channels)), n_out), dtype=np.complex128)
    for ci, k in enumerate(channels):
        xs = np.concatenate([np.zeros(L - 1, dtype=np.complex128), x * np.exp(-2j * np.pi * ((k * np.arange(N)) % M) / M)])
        for m in range(n_out):
            seg = xs[m * D:m * D + L]          # x_shifted[mD-(L-1) .. mD]
            out[ci, m] = np.dot(seg, h[::-1])
    return out
