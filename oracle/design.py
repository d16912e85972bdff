"""ORACLE (test infrastructure, not product code): numpy restatement of the
scipy.signal filter-design calls the reference makes on the hot path.

The arithmetic of the reference's decimator and channel filter lives in a
third-party dependency, scipy (requirements.txt:2 `scipy>=1.10.0`, unpinned;
pinned here to scipy 1.15.3 / numpy 2.2.6, the versions the goldens were made
with).  Call sites: tetraear/signal/processor.py:254 (`signal.decimate` ->
`cheby1(8, 0.05, 0.8/q, output='sos')`, `sosfilt_zi`) and :78 (`signal.butter(4,
cutoff)`; `filtfilt` -> `lfilter_zi`).  Each function below follows the published
scipy algorithm (scipy/signal/_filter_design.py: cheb1ap, buttap, lp2lp_zpk,
bilinear_zpk, zpk2sos, zpk2tf; _signaltools.py: lfilter_zi, sosfilt_zi) and is
pinned against tables dumped from scipy itself (tests/golden/design.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np


def _cheb1ap(N, rp):
    eps = np.sqrt(10 ** (0.1 * rp) - 1.0)
    mu = 1.0 / N * np.arcsinh(1 / eps)
    m = np.arange(-N + 1, N, 2)
    theta = np.pi * m / (2 * N)
    p = -np.sinh(mu + 1j * theta)
    k = np.prod(-p, axis=0).real
    if N % 2 == 0:
        k = k / np.sqrt(1 + eps * eps)
    return p, k


def _buttap(N):
    m = np.arange(-N + 1, N, 2)
    p = -np.exp(1j * np.pi * m / (2 * N))
    return p, 1.0


def _lowpass_digital_zpk(p, k, Wn):
    """prewarp + lp2lp_zpk + bilinear_zpk (fs=2) for an all-pole prototype."""
    fs = 2.0
    warped = 2 * fs * np.tan(np.pi * Wn / fs)
    degree = len(p)
    p_lp = warped * p
    k_lp = k * warped ** degree
    fs2 = 2.0 * fs
    p_z = (fs2 + p_lp) / (fs2 - p_lp)
    z_z = -np.ones(degree)
    k_z = k_lp * np.real(1.0 / np.prod(fs2 - p_lp))
    return z_z, p_z, k_z


def _pair_conj(p):
    """_cplxreal for an all-complex, conjugate-symmetric pole set: one pole per pair
    (positive imaginary part), averaged with its conjugate partner, sorted by real part."""
    p = p[np.lexsort((np.abs(p.imag), p.real))]
    zp = p[p.imag > 0]
    zn = p[p.imag < 0]
    assert len(zp) == len(zn) and len(zp) * 2 == len(p)
    return (zp + zn.conj()) / 2


def cheby1_lowpass_sos(N, rp, Wn):
    """cheby1(N, rp, Wn, output='sos') for even N (all poles complex)."""
    assert N % 2 == 0
    p, k = _cheb1ap(N, rp)
    z, p, k = _lowpass_digital_zpk(p, k, Wn)
    pc = _pair_conj(p)
    nsec = N // 2
    sos = np.zeros((nsec, 6))
    # zpk2sos, pairing 'nearest': sections are filled last-to-first with the pole
    # closest to the unit circle first; all zeros are at -1 so each gets (z+1)^2.
    for si in range(nsec - 1, -1, -1):
        idx = np.argmin(np.abs(1 - np.abs(pc)))
        p1 = pc[idx]
        pc = np.delete(pc, idx)
        a = np.poly([p1, p1.conj()]).real
        b = np.poly([-1.0, -1.0])
        sos[si, :3] = b
        sos[si, 3:] = a
    sos[0, :3] *= k
    return sos


def butter_lowpass_ba(N, Wn):
    """butter(N, Wn, btype='low') -> (b, a)."""
    p, k = _buttap(N)
    z, p, k = _lowpass_digital_zpk(p, k, Wn)
    b = k * np.poly(z)
    a = np.poly(p)
    return np.real(b), np.real(a)


def lfilter_zi(b, a):
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if a[0] != 1.0:
        b = b / a[0]
        a = a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]
    b = np.r_[b, np.zeros(n - len(b))]
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    for i in range(1, n - 1):
        comp[i, i - 1] = 1.0
    IminusA = np.eye(n - 1) - comp.T
    B = b[1:] - a[1:] * b[0]
    return np.linalg.solve(IminusA, B)


def sosfilt_zi(sos):
    sos = np.asarray(sos, dtype=np.float64)
    zi = np.empty((sos.shape[0], 2))
    scale = 1.0
    for s in range(sos.shape[0]):
        b = sos[s, :3]
        a = sos[s, 3:]
        zi[s] = scale * lfilter_zi(b, a)
        scale *= b.sum() / a.sum()
    return zi


class RateParams:
    """Derived constants of process() for one sample rate (processor.py:245-255,
    :69-75, :183, :194)."""

    def __init__(self, sample_rate):
        self.sample_rate = float(sample_rate)
        self.q = 1
        if self.sample_rate > 240000 * 2:
            q = int(self.sample_rate / 240000)
            if q > 1:
                self.q = q
        self.rate_dec = self.sample_rate / self.q if self.q > 1 else self.sample_rate
        if self.q > 1:
            self.sos = cheby1_lowpass_sos(8, 0.05, 0.8 / self.q)
            self.soszi = sosfilt_zi(self.sos)
        else:
            self.sos = None
            self.soszi = None


def butter_for(bandwidth, fs):
    """The design filter_signal() performs (processor.py:69-78)."""
    nyquist = fs / 2
    cutoff = (bandwidth / 2) / nyquist
    cutoff = min(0.99, max(0.01, cutoff))
    b, a = butter_lowpass_ba(4, cutoff)
    return b, a, lfilter_zi(b, a)
