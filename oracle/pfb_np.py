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


OCC_FFT = 256


def occupancy(rows, M, chan_rate, snr_db=15.0, min_dbfs=-70.0, peak_db=3.0, bandwidth_hz=25000.0):
    """Occupancy gate of a channeliser's output (rows: [streams * M][>= 256] complex): the reference's gate
    (tetraear/ui/modern.py:1921-2003) per channel row -- Hann-windowed FFT of the row's first 256 samples, power in dBFS,
    mean and peak over the bins within 25 kHz around the centre, strong = snr > 15 and peak > -70 and peak - mean > 3 --
    with the noise floor taken as the MEDIAN of the channels' mean in-band power per stream (the reference's floor, the
    bins outside the centre channel, would count neighbouring carriers as noise).
    Returns (signal_power[rows], peak_power[rows], occupied[rows] bool)."""
    rows = np.asarray(rows)
    n_fft = OCC_FFT
    x = rows[:, :n_fft].astype(np.complex128)
    window = np.hanning(n_fft)
    fft = np.fft.fftshift(np.fft.fft(x * window, axis=1), axes=1)
    power = 20 * np.log10(np.abs(fft) / n_fft + 1e-20)
    center_idx = n_fft // 2
    bandwidth_bins = int(bandwidth_hz / (chan_rate / n_fft))
    start_idx = max(0, center_idx - bandwidth_bins // 2)
    end_idx = min(n_fft, center_idx + bandwidth_bins // 2)
    signal_power = power[:, start_idx:end_idx].mean(axis=1)
    peak_power = power[:, start_idx:end_idx].max(axis=1)
    floor = np.repeat(np.median(signal_power.reshape(-1, M), axis=1), M)
    occupied = (signal_power - floor > snr_db) & (peak_power > min_dbfs) & (peak_power - signal_power > peak_db)
    return signal_power, peak_power, occupied


def occupancy_bins(chan_rate, bandwidth_hz=25000.0):
    """[start_idx, end_idx) of the in-band bins (what the library's kernel is told)"""
    bandwidth_bins = int(bandwidth_hz / (chan_rate / OCC_FFT))
    return max(0, OCC_FFT // 2 - bandwidth_bins // 2), min(OCC_FFT, OCC_FFT // 2 + bandwidth_bins // 2)
