"""Spectrum / AFC / signal gate on the GPU: the decision
`CaptureThread.run` makes before calling process() (tetraear/ui/modern.py:1921-2021)."""
import numpy as np

from tetraear_amd import _lib
from tetraear_amd._lib import FMT_BYTES, check, ptr

_FMT_OF = {"cu8": 0, "cs8": 1, "cf32": 2, "cf64": 3}
FIELDS = ("peak_freq_offset", "signal_power", "peak_power", "noise_floor", "snr", "strong", "afc")


def spectrum_gate(iq, fmt, n_samples, rows=1, sample_rate=2.4e6, device=0):
    """iq: `rows` streams of n_samples back to back.  Returns (list of dicts per row, afc array)."""
    f = _FMT_OF[fmt]
    iq = np.ascontiguousarray(iq)
    assert iq.nbytes >= rows * n_samples * FMT_BYTES[f]
    out = np.zeros((rows, 8))
    afc = np.zeros(rows)
    check(_lib.load().tdm_spectrum_gate(ptr(iq), f, n_samples, n_samples, rows, float(sample_rate), ptr(out), ptr(afc),
                                        0, device))
    res = []
    for r in range(rows):
        d = {k: float(out[r, i]) for i, k in enumerate(FIELDS)}
        d["strong"] = bool(out[r, 5] != 0.0)
        res.append(d)
    return res, afc


def occupancy_gate(chan, M, chan_rate, snr_db=15.0, min_dbfs=-70.0, device=0):
    """Many-carrier form of the gate: which rows of a channeliser's output carry a signal (tdm_occupancy_gate; the
    reference's rule of ui/modern.py:1921-2003 per channel row, noise floor = median over the stream's channels).
    chan: complex64 [streams * M][n_out >= 256].  Returns (signal_power dBFS, peak_power dBFS, occupied bool, row list)."""
    chan = np.ascontiguousarray(chan, dtype=np.complex64)
    rows, n_out = chan.shape
    assert rows % M == 0
    stats = np.zeros((rows, 2), dtype=np.float32)
    flags = np.zeros(rows, dtype=np.uint8)
    row_list = np.zeros(rows, dtype=np.int32)
    n_rows = np.zeros(1, dtype=np.int32)
    check(_lib.load().tdm_occupancy_gate(ptr(chan), n_out, rows // M, M, n_out, float(chan_rate), float(snr_db), float(min_dbfs),
                                         ptr(stats), ptr(flags), ptr(row_list), ptr(n_rows), None, 0, device))
    return stats[:, 0], stats[:, 1], flags.astype(bool), np.sort(row_list[:int(n_rows[0])])
