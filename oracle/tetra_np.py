"""ORACLE for TETRA mode (test infrastructure): fp64 numpy restatement of the tetra-mode receiver.

TETRA mode has NO reference implementation: syrex1013/TetraEar contains no RRC filter, no timing
recovery and no pi/4-DQPSK quadrant slicer.  This file therefore pins nothing
against the reference ("parity unpinned" for this mode); it is the project's own definition of
the algorithm, written in plain numpy/fp64, against which the fp32 HIP kernels are checked, and
which is itself checked against the known transmitted symbols of the synthetic generator.

Algorithm (per carrier, per chunk, stateless like the reference's process()):
  1. matched filter: centred root-raised-cosine FIR (alpha 0.35, span 8 symbols, unit energy),
     zero-padded at the chunk edges; output at the input rate.
  2. timing: feed-forward square-law (Oerder-Meyr) estimate per sub-block of `TB` samples,
     C_b = sum |y[n]|^2 exp(-2 pi i n / sps); vector-averaged over +-`TW` sub-blocks with prefix
     sums; tau_b = -arg(C_b)/(2 pi) unwrapped along the chunk; linear interpolation between
     sub-block centres gives tau(k) for symbol k.
  3. symbol instants t_k = (k + tau(k)) * sps; cubic Lagrange (Farrow) interpolation of y.
  4. differential detection d_k = s_k conj(s_{k-1}); residual carrier offset from the 4th-power
     estimate delta = arg(-sum d_k^4)/4; decision = quadrant of d_k exp(-i delta):
     +pi/4 -> 0, +3pi/4 -> 1, -pi/4 -> 2, -3pi/4 -> 3 (ETSI EN 300 392-2 table 5.1 as quoted in
     tetraear/signal/processor.py:106-110).
"""
import numpy as np

SYMBOL_RATE = 18000.0
ALPHA = 0.35
SPAN = 8      # symbols
TB = 256      # samples per timing sub-block
TW = 2        # sub-blocks averaged on each side


def rrc_taps(sps, alpha=ALPHA, span=SPAN, exact=False):
    """Centred RRC taps, odd length, unit energy.  exact=False: the 16-bit coefficients the device's plan designs
    (coeff16 below); exact=True: the same taps in full float64, the UNQUANTISED filter the soft-symbol tolerance of
    BASELINE.json's north_star (1e-5) is also asserted against (tests/test_tetra_precision.py)."""
    half = int(np.floor(span * sps / 2))
    t = np.arange(-half, half + 1, dtype=np.float64) / sps
    h = np.empty_like(t)
    eps = 1e-9
    for i, ti in enumerate(t):
        if abs(ti) < eps:
            h[i] = 1.0 - alpha + 4 * alpha / np.pi
        elif abs(abs(ti) - 1.0 / (4 * alpha)) < eps:
            h[i] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                           + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            h[i] = (np.sin(np.pi * ti * (1 - alpha)) + 4 * alpha * ti * np.cos(np.pi * ti * (1 + alpha))) \
                / (np.pi * ti * (1 - (4 * alpha * ti) ** 2))
    h = h / np.sqrt(np.sum(h * h))
    return h if exact else coeff16(h)


def _bf16(x):
    """float32 -> nearest bfloat16 (round to nearest even), returned as float32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def coeff16(h):
    """The matched filter's coefficients have 16 significant bits: each is the sum of two bfloat16 (leading part +
    rounded remainder).  The device multiplies on the bf16 matrix cores with samples and coefficients split that way;
    coefficients that ARE such sums leave no coefficient rounding in its products (the change to the filter is at the
    -96 dB level)."""
    f = np.asarray(h, dtype=np.float32)
    hi = _bf16(f)
    lo = _bf16(f - hi)
    return hi.astype(np.float64) + lo.astype(np.float64)


def matched_filter(x, h):
    """y[n] = sum_t h[t] x[n + t - (T-1)/2], zero outside the chunk (h is symmetric)."""
    return np.convolve(x, h, mode="same")


def timing_estimates(y, sps):
    n = len(y)
    nb = (n + TB - 1) // TB
    idx = np.arange(n, dtype=np.float64)
    w = (np.abs(y) ** 2) * np.exp(-2j * np.pi * idx / sps)
    C = np.array([np.sum(w[b * TB:(b + 1) * TB]) for b in range(nb)])
    P = np.concatenate([[0], np.cumsum(C)])
    Cs = np.array([P[min(nb, b + TW + 1)] - P[max(0, b - TW)] for b in range(nb)])
    tau = -np.angle(Cs) / (2 * np.pi)
    # unwrap in units of one symbol
    out = np.empty(nb)
    prev = 0.0
    for b in range(nb):
        t = tau[b]
        if b > 0:
            t += np.round(prev - t)
        out[b] = t
        prev = t
    return out


def tau_of_sample(pos, tau_b):
    """piecewise-linear tau at sample position pos (sub-block centres at (b+0.5)*TB)."""
    nb = len(tau_b)
    u = pos / TB - 0.5
    b0 = np.clip(np.floor(u).astype(np.int64), 0, max(nb - 2, 0))
    if nb == 1:
        return np.full_like(np.asarray(pos, dtype=np.float64), tau_b[0])
    f = np.clip(u - b0, 0.0, 1.0)
    return tau_b[b0] * (1 - f) + tau_b[b0 + 1] * f


def farrow(y, t):
    """cubic Lagrange interpolation of y at fractional positions t (1 <= t <= len-3)."""
    m = np.floor(t).astype(np.int64)
    mu = t - m
    ym1, y0, y1, y2 = y[m - 1], y[m], y[m + 1], y[m + 2]
    c0 = y0
    c1 = y1 - ym1 / 3 - y0 / 2 - y2 / 6
    c2 = (ym1 + y1) / 2 - y0
    c3 = (y2 - ym1) / 6 + (y0 - y1) / 2
    return ((c3 * mu + c2) * mu + c1) * mu + c0


def demod(x, sample_rate, exact_taps=False):
    """Returns (hard uint8[n_sym-1], soft complex[n_sym-1] = derotated d_k, info dict).
    exact_taps=True runs the matched filter with the unquantised float64 RRC (see rrc_taps)."""
    x = np.asarray(x, dtype=np.complex128)
    sps = sample_rate / SYMBOL_RATE
    h = rrc_taps(sps, exact=exact_taps)
    y = matched_filter(x, h)
    n = len(y)
    tau_b = timing_estimates(y, sps)
    # symbol instants: k such that 1 <= t_k <= n-3 using the nominal position k*sps for tau lookup
    kmax = int(np.floor(n / sps)) + 2
    k = np.arange(0, kmax, dtype=np.float64)
    t = (k + tau_of_sample(k * sps, tau_b)) * sps
    ok = (t >= 1.0) & (t <= n - 3.0)
    t = t[ok]
    s = farrow(y, t)
    d = s[1:] * np.conj(s[:-1])
    if len(d) == 0:
        return np.zeros(0, np.uint8), d, dict(tau=tau_b, delta=0.0, n_sym=len(s), sym=s, t=t)
    acc = np.sum(d ** 4)
    delta = np.angle(-acc) / 4 if acc != 0 else 0.0
    dd = d * np.exp(-1j * delta)
    hard = np.where(dd.imag >= 0, np.where(dd.real >= 0, 0, 1), np.where(dd.real >= 0, 2, 3)).astype(np.uint8)
    margin = np.min(np.minimum(np.arctan2(np.abs(dd.imag), np.abs(dd.real)),
                               np.pi / 2 - np.arctan2(np.abs(dd.imag), np.abs(dd.real))))
    return hard, dd, dict(tau=tau_b, delta=delta, n_sym=len(s), t=t, sym=s, margin=margin)


# ------------------------------------------------------------------------------------------------------------------
# Comparison receiver: the textbook feedback loop BASELINE.json's north_star names (Gardner timing-error detector +
# proportional-integral loop filter + Farrow interpolator), fp64, strictly sequential.  It is NOT what the device
# runs -- the device runs the feed-forward estimator above, which has no recurrence over symbols and is therefore
# fully parallel -- it exists so that tests/test_tetra_gardner.py can show, point by point over Es/N0, timing offset
# and carrier offset, that the substitution costs nothing in symbol error rate or timing jitter.
# ------------------------------------------------------------------------------------------------------------------
def _farrow1(y, t):
    m = int(np.floor(t))
    mu = t - m
    ym1, y0, y1, y2 = y[m - 1], y[m], y[m + 1], y[m + 2]
    c1 = y1 - ym1 / 3 - y0 / 2 - y2 / 6
    c2 = (ym1 + y1) / 2 - y0
    c3 = (y2 - ym1) / 6 + (y0 - y1) / 2
    return ((c3 * mu + c2) * mu + c1) * mu + y0


GARDNER_WARMUP_SYMBOLS = 512


def gardner_segments(n, sample_rate, ntaps=None):
    """Geometry of the two-halves form of the Gardner receiver (the library's rule, tdm_hip.hip): each half is n_v samples
    long -- half the chunk plus an overlap that holds 512 warm-up symbols of the second half's loop -- the second starts
    seg_off samples into the chunk, and the seam lies `margin` samples before the first half's end (clear of its matched
    filter's edge).  Returns None when the chunk is too short for two halves (n_v + 8 overlaps > n)."""
    sps = sample_rate / SYMBOL_RATE
    if ntaps is None:
        ntaps = len(rrc_taps(sps))
    margin = (ntaps - 1) // 2 + 4 * int(np.ceil(sps)) + 8
    ov = ((int(np.ceil(GARDNER_WARMUP_SYMBOLS * sps)) + margin + 1) // 2 + 1) & ~1
    n_v = ((n // 2 + ov) + 1) & ~1
    if n_v + 8 * ov > n:
        return None
    seg_off = n - n_v
    return dict(n_v=n_v, seg_off=seg_off, seam_a=n_v - margin, seam_b=n_v - margin - seg_off)


def _gardner_loop(x, sample_rate, bn_t, zeta):
    """one loop over one stretch of samples: symbols and their instants"""
    sps = sample_rate / SYMBOL_RATE
    y = matched_filter(x, rrc_taps(sps))
    n = len(y)
    # loop constants (Rice, Digital Communications, eq. C.61) for detector gain kp (S-curve slope of the
    # normalised Gardner detector for RRC alpha 0.35, about 2.7 per symbol) and unit NCO gain
    kp = 2.7
    th = bn_t / (zeta + 0.25 / zeta)
    k1 = 4 * zeta * th / (1 + 2 * zeta * th + th * th) / kp
    k2 = 4 * th * th / (1 + 2 * zeta * th + th * th) / kp
    t = 1.0 + sps          # first symbol instant (the loop pulls it onto the eye)
    integ = 0.0
    pw = 1.0               # running symbol power
    ts, s = [], []
    prev = None
    while t <= n - 3.0:
        sk = _farrow1(y, t)
        if prev is not None:
            mid = _farrow1(y, t - 0.5 * sps * (1.0 - integ))
            pw = 0.99 * pw + 0.01 * (abs(sk) ** 2)
            e = ((sk - prev) * np.conj(mid)).real / max(pw, 1e-12)
            integ += k2 * e
            v = k1 * e + integ
        else:
            v = 0.0
        ts.append(t)
        s.append(sk)
        prev = sk
        t += sps * (1.0 - v)     # (a late strobe makes e positive: shorten the period)
    return np.array(s), np.array(ts)


def demod_gardner(x, sample_rate, bn_t=0.01, zeta=0.7071, segments=1):
    """Gardner TED (e_k = Re{(s_k - s_{k-1}) conj(s_{k-1/2})}, normalised by the running symbol power) -> PI loop
    (noise bandwidth bn_t symbol rates, damping zeta) -> period-controlled Farrow interpolation of the matched-filter
    output; then the same differential detection, 4th-power carrier-offset estimate and quadrant slicer as demod().
    Returns (hard, derotated d_k, info with the symbol instants `t`).

    segments = 2 (what the library does for batches that would leave most of the device idle, tdm_plan_info.gardner_segments):
    the chunk as TWO independently started loops over samples [0, n_v) and [n - n_v, n) (gardner_segments: n_v = half the
    chunk plus an overlap of 512 warm-up symbols), joined at a seam near the overlap's end: the first loop's symbols whose
    instant's whole part lies before the seam, then the second loop's from the symbol that is the first loop's first one at
    or behind the seam (their instants relative to the seam differ by a whole number of symbol periods, 0 unless the two
    loops place a symbol on different sides of the seam).  The loop is a contraction, so the second loop runs onto the
    first one's trajectory during its warm-up; the chunk's second half starts as every chunk does -- stateless."""
    x = np.asarray(x, dtype=np.complex128)
    sps = sample_rate / SYMBOL_RATE
    geo = gardner_segments(len(x), sample_rate) if segments == 2 else None
    if geo is None:
        s, ts = _gardner_loop(x, sample_rate, bn_t, zeta)
    else:
        sa, ta = _gardner_loop(x[:geo["n_v"]], sample_rate, bn_t, zeta)
        sb, tb = _gardner_loop(x[geo["seg_off"]:], sample_rate, bn_t, zeta)
        ia = np.nonzero(np.floor(ta) >= geo["seam_a"])[0]
        ib = np.nonzero(np.floor(tb) >= geo["seam_b"])[0]
        ka = int(ia[0]) if len(ia) else len(sa)
        jb = int(ib[0]) if len(ib) else len(sb)
        rel_a = (ta[ka] - geo["seam_a"]) if ka < len(sa) else 0.0
        rel_b = (tb[jb] - geo["seam_b"]) if jb < len(sb) else 0.0
        d = int(np.rint(np.float32(rel_b - rel_a) / np.float32(sps)))
        jb = min(max(jb - d, 0), len(sb))
        s = np.concatenate([sa[:ka], sb[jb:]])
        ts = np.concatenate([ta[:ka], tb[jb:] + geo["seg_off"]])
    d = s[1:] * np.conj(s[:-1])
    if len(d) == 0:
        return np.zeros(0, np.uint8), d, dict(t=ts)
    acc = np.sum(d ** 4)
    delta = np.angle(-acc) / 4 if acc != 0 else 0.0
    dd = d * np.exp(-1j * delta)
    hard = np.where(dd.imag >= 0, np.where(dd.real >= 0, 0, 1), np.where(dd.real >= 0, 2, 3)).astype(np.uint8)
    return hard, dd, dict(t=ts, delta=delta)
