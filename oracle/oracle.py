"""ORACLE (test infrastructure): Python face of oracle/ref_chain.c.

`OracleSignalProcessor` mirrors the reference class
`tetraear.signal.processor.SignalProcessor` (processor.py:18-273) method for
method so parity tests read like the reference's own tests.  Arithmetic is done
by liboracle.so (plain C, single thread); filter design by oracle/design.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import design

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_decimate.restype = C.c_int64
        L.orc_decimate.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int64, _dp, _dp]
        L.orc_frequency_shift.restype = None
        L.orc_frequency_shift.argtypes = [_dp, _dp, C.c_int64, C.c_double, C.c_double]
        L.orc_filtfilt.restype = C.c_int
        L.orc_filtfilt.argtypes = [_dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int64]
        L.orc_extract_symbols.restype = C.c_int64
        L.orc_extract_symbols.argtypes = [_dp, _dp, C.c_int64, C.c_double, C.c_double, _dp, _dp,
                                          C.POINTER(C.c_int32), _dp]
        L.orc_demodulate_dqpsk.restype = C.c_int64
        L.orc_demodulate_dqpsk.argtypes = [_dp, _dp, C.c_int64, _u8p, _dp]
        L.orc_process.restype = C.c_int64
        L.orc_process.argtypes = [_dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_int, _dp, _dp, C.c_int,
                                  _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int64), _u8p,
                                  C.POINTER(C.c_int32), _dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _split(x):
    x = np.asarray(x)
    xr = np.ascontiguousarray(x.real, dtype=np.float64)
    xi = np.ascontiguousarray(x.imag, dtype=np.float64) if np.iscomplexobj(x) else np.zeros(len(xr))
    return xr, xi


def _join(re, im):
    """re + 1j*im without the arithmetic (`1j * inf` is `nan + inf j`): the components as they are"""
    out = np.empty(len(re), dtype=np.complex128)
    out.real, out.imag = re, im
    return out


_DESIGN_CACHE = {}


def _rate_design(sample_rate):
    key = float(sample_rate)
    if key not in _DESIGN_CACHE:
        rp = design.RateParams(sample_rate)
        b1, a1, zi1 = design.butter_for(25000, rp.rate_dec)
        b0, a0, zi0 = design.butter_for(25000, rp.sample_rate)
        _DESIGN_CACHE[key] = (rp, (b1, a1, zi1), (b0, a0, zi0))
    return _DESIGN_CACHE[key]


class OracleSignalProcessor:
    """CPU oracle with the reference's interface (processor.py:18)."""

    def __init__(self, sample_rate=2.4e6):
        self.sample_rate = sample_rate
        self.symbol_rate = 18000
        self.samples_per_symbol = int(sample_rate / self.symbol_rate)
        self.symbols = None
        self.best_phase = None
        self.min_margin = None

    # processor.py:51-83
    def filter_signal(self, samples, bandwidth=25000, sample_rate=None):
        if len(samples) == 0:
            return samples
        fs = sample_rate if sample_rate is not None else self.sample_rate
        b, a, zi = design.butter_for(bandwidth, fs)
        xr, xi = _split(samples)
        rc = lib().orc_filtfilt(_p(b), _p(a), _p(zi), 4, _p(xr), _p(xi), len(xr))
        if rc != 0:
            return samples
        return _join(xr, xi)

    # processor.py:85-100
    def frequency_shift(self, samples, freq_offset, sample_rate=None):
        fs = sample_rate if sample_rate is not None else self.sample_rate
        xr, xi = _split(samples)
        lib().orc_frequency_shift(_p(xr), _p(xi), len(xr), float(freq_offset), float(fs))
        return _join(xr, xi)

    # processor.py:102-166
    def demodulate_dqpsk(self, samples):
        if len(samples) < 2:
            return np.array([], dtype=np.uint8)
        xr, xi = _split(samples)
        out = np.empty(len(xr) - 1, dtype=np.uint8)
        margin = C.c_double()
        lib().orc_demodulate_dqpsk(_p(xr), _p(xi), len(xr), out.ctypes.data_as(_u8p), C.byref(margin))
        self.min_margin = margin.value
        return out

    # processor.py:168-219
    def extract_symbols(self, samples, sample_rate=None, return_powers=False):
        if len(samples) == 0:
            return np.array([], dtype=complex)
        fs = sample_rate if sample_rate is not None else self.sample_rate
        xr, xi = _split(samples)
        sr = np.empty(len(xr))
        si = np.empty(len(xr))
        bp = C.c_int32()
        sps = max(1, int(fs / self.symbol_rate))
        powers = np.full(sps, -1.0)
        n = lib().orc_extract_symbols(_p(xr), _p(xi), len(xr), float(fs), float(self.symbol_rate), _p(sr),
                                      _p(si), C.byref(bp), _p(powers))
        self.best_phase = bp.value
        out = _join(sr[:n], si[:n])
        if return_powers:
            return out, powers
        return out

    def decimate(self, samples, q):
        """scipy.signal.decimate(samples, q) as called at processor.py:254."""
        sos = design.cheby1_lowpass_sos(8, 0.05, 0.8 / q)
        zi = design.sosfilt_zi(sos)
        xr, xi = _split(samples)
        m = (len(xr) + q - 1) // q
        yr = np.empty(max(m, 1))
        yi = np.empty(max(m, 1))
        r = lib().orc_decimate(_p(sos), _p(zi), sos.shape[0], q, _p(xr), _p(xi), len(xr), _p(yr), _p(yi))
        if r < 0:
            raise ValueError("The length of the input vector x must be greater than padlen, which is 27.")
        return _join(yr[:r], yi[:r])

    # processor.py:221-273
    def process(self, samples, freq_offset=0):
        if len(samples) == 0:
            self.symbols = np.array([], dtype=complex)
            return np.array([], dtype=np.uint8)
        rp, (b1, a1, zi1), (b0, a0, zi0) = _rate_design(self.sample_rate)
        xr, xi = _split(samples)
        n = len(xr)
        sr = np.empty(n)
        si = np.empty(n)
        hard = np.empty(n, dtype=np.uint8)
        ns = C.c_int64()
        bp = C.c_int32()
        margin = C.c_double()
        sos = rp.sos if rp.q > 1 else np.zeros((4, 6))
        soszi = rp.soszi if rp.q > 1 else np.zeros((4, 2))
        nh = lib().orc_process(_p(xr), _p(xi), n, float(self.sample_rate), float(freq_offset), rp.q,
                               _p(sos), _p(soszi), 4, _p(b1), _p(a1), _p(zi1), _p(b0), _p(a0), _p(zi0),
                               _p(sr), _p(si), C.byref(ns), hard.ctypes.data_as(_u8p), C.byref(bp),
                               C.byref(margin))
        self.symbols = _join(sr[:ns.value], si[:ns.value])
        self.best_phase = bp.value
        self.min_margin = margin.value
        return hard[:nh].copy()


def resample_np(samples, sample_rate, target_rate):
    """scipy.signal.resample as called at processor.py:46-48 (FFT method, complex input):
    X = fft(x); keep the N = min(num, Nx) lowest-|f| bins; y = ifft(Y) * num/Nx."""
    x = np.asarray(samples)
    Nx = len(x)
    num = int(Nx * target_rate / sample_rate)
    X = np.fft.fft(x)
    Y = np.zeros(num, dtype=X.dtype)
    N = min(num, Nx)
    nyq = N // 2 + 1
    Y[:nyq] = X[:nyq]
    if N > 2:
        Y[nyq - N:] = X[nyq - N:]
    if N % 2 == 0:
        if num < Nx:  # downsampling: fold the Nyquist bin
            sl = slice(-N // 2, -N // 2 + 1)  # scipy's own slice form: empty when N == 2
            Y[sl] += X[sl]
        elif Nx < num:  # upsampling: split it
            Y[N // 2] *= 0.5
            Y[num - N // 2] = Y[N // 2]
    y = np.fft.ifft(Y)
    y *= float(num) / float(Nx)
    return y
