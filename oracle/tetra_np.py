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


GARDNER_WARMUP_SYMBOLS = 384
GARDNER_INIT_SAMPLES = 512      # what a piece's first instant is estimated from (the device's ring of filter outputs)


GARDNER_MAX_PIECES = 8


def gardner_segments(n, sample_rate, ntaps=None, pieces=2):
    """Geometry of the Gardner receiver run as `pieces` independently started loops per chunk (the library's, tdm_hip.hip):
    every piece is n_v samples long, piece p starts p * seg_step samples into the chunk (the last one ends with it); the
    seam between pieces p and p + 1 lies `margin` samples before piece p's end (clear of its matched filter's edge), and
    n_v is such that a piece has run for GARDNER_WARMUP_SYMBOLS (384) when it reaches the seam it takes over at:
    in a piece's own coordinates  seam_out = n_v - margin  (none for the last piece),  seam_in = seam_out - seg_step  (none
    for the first).  Returns None when the chunk is too short (a piece's own part, n / pieces, shorter than 1.9 warm-ups)."""
    sps = sample_rate / SYMBOL_RATE
    if ntaps is None:
        ntaps = len(rrc_taps(sps))
    K = int(pieces)
    if K < 2:
        return None
    margin = (ntaps - 1) // 2 + 4 * int(np.ceil(sps)) + 8
    lead = int(np.ceil(GARDNER_WARMUP_SYMBOLS * sps)) + margin
    if 10 * n < 19 * K * lead:
        return None
    n_v0 = (n + (K - 1) * lead + K - 1) // K
    seg_step = (n - n_v0) // (K - 1)
    n_v = n - (K - 1) * seg_step
    return dict(pieces=K, n_v=n_v, seg_step=seg_step, seam_out=n_v - margin, seam_in=n_v - margin - seg_step, margin=margin)


def _gardner_loop(x, sample_rate, bn_t, zeta, ff_init=False):
    """one loop over one stretch of samples: symbols and their instants.
    ff_init (the pieces of a chunk behind the first, demod_gardner(segments=K)): the first instant is not 1 + sps but the
    instant in [1 + sps, 1 + 2 sps) that the square-law estimate over the first 512 filter outputs puts a symbol at --
    the maximum of the symbol-rate component of |y|^2, tau = -arg(sum_n |y_n|^2 exp(-2 pi i n / sps)) sps / (2 pi) -- so that
    a piece's loop starts next to the eye instead of wherever its first sample happens to lie: started half a symbol off,
    the Gardner detector's error is zero too, and the loop can sit there for hundreds of symbols before it pulls in."""
    sps = sample_rate / SYMBOL_RATE
    taps = rrc_taps(sps)
    y = matched_filter(x, taps)
    n = len(y)
    # loop constants (Rice, Digital Communications, eq. C.61) for detector gain kp (S-curve slope of the
    # normalised Gardner detector for RRC alpha 0.35, about 2.7 per symbol) and unit NCO gain
    kp = 2.7
    th = bn_t / (zeta + 0.25 / zeta)
    k1 = 4 * zeta * th / (1 + 2 * zeta * th + th * th) / kp
    k2 = 4 * th * th / (1 + 2 * zeta * th + th * th) / kp
    t = 1.0 + sps          # first symbol instant (the loop pulls it onto the eye)
    if ff_init and n >= GARDNER_INIT_SAMPLES:
        h = (len(taps) - 1) // 2
        nn = np.arange(h, GARDNER_INIT_SAMPLES)
        c = np.sum(np.abs(y[h:GARDNER_INIT_SAMPLES]) ** 2 * np.exp(-2j * np.pi * nn / sps))
        tau = -np.angle(c) * sps / (2 * np.pi)
        t = t + (tau - t) % sps
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


def demod_gardner(x, sample_rate, bn_t=0.01, zeta=0.7071, segments=1, ff_first=False):
    """Gardner TED (e_k = Re{(s_k - s_{k-1}) conj(s_{k-1/2})}, normalised by the running symbol power) -> PI loop
    (noise bandwidth bn_t symbol rates, damping zeta) -> period-controlled Farrow interpolation of the matched-filter
    output; then the same differential detection, 4th-power carrier-offset estimate and quadrant slicer as demod().
    Returns (hard, derotated d_k, info with the symbol instants `t`).

    segments = K >= 2 (what the library does for batches that would leave most of the device idle, tdm_plan_info.gardner_segments):
    the chunk as K independently started loops over the pieces of gardner_segments(pieces=K), joined at the K - 1 seams: a
    piece's symbols up to its first one at or behind its outgoing seam, then the next piece's from the symbol that IS that
    one (the two loops' instants relative to the seam differ by a whole number of symbol periods, 0 unless they place a
    symbol on different sides of the seam).  The loop is a contraction, so a piece's loop -- started next to the eye
    (_gardner_loop ff_init) -- runs onto its predecessor's trajectory during its 384 warm-up symbols; stateless like a chunk.

    ff_first (the library's plan option "gardner_ff_start"): the chunk's FIRST loop -- the only one of a whole chunk --
    starts at the feed-forward estimate too, instead of at 1 + sps: no hang-up at the start of a chunk either."""
    x = np.asarray(x, dtype=np.complex128)
    sps = sample_rate / SYMBOL_RATE
    geo = gardner_segments(len(x), sample_rate, pieces=segments) if segments >= 2 else None
    if geo is None:
        s, ts = _gardner_loop(x, sample_rate, bn_t, zeta, ff_init=ff_first)
    else:
        K, n_v, step = geo["pieces"], geo["n_v"], geo["seg_step"]
        s_parts, t_parts = [], []
        start = 0                 # index of the piece's first kept symbol
        t_out_prev = 0.0
        for p in range(K):
            sp, tp = _gardner_loop(x[p * step:p * step + n_v], sample_rate, bn_t, zeta, ff_init=p > 0 or ff_first)
            if p > 0:
                i_in = np.nonzero(np.floor(tp) >= geo["seam_in"])[0]
                k_in = int(i_in[0]) if len(i_in) else len(sp)
                rel_in = (tp[k_in] - geo["seam_in"]) if k_in < len(sp) else 0.0
                d = int(np.rint(np.float32(rel_in - t_out_prev) / np.float32(sps)))
                start = min(max(k_in - d, 0), len(sp))
            end = len(sp)
            if p < K - 1:
                i_out = np.nonzero(np.floor(tp) >= geo["seam_out"])[0]
                k_out = int(i_out[0]) if len(i_out) else len(sp)
                t_out_prev = (tp[k_out] - geo["seam_out"]) if k_out < len(sp) else 0.0
                end = k_out
            end = max(end, start)
            s_parts.append(sp[start:end])
            t_parts.append(tp[start:end] + p * step)
        s = np.concatenate(s_parts)
        ts = np.concatenate(t_parts)
    d = s[1:] * np.conj(s[:-1])
    if len(d) == 0:
        return np.zeros(0, np.uint8), d, dict(t=ts)
    acc = np.sum(d ** 4)
    delta = np.angle(-acc) / 4 if acc != 0 else 0.0
    dd = d * np.exp(-1j * delta)
    hard = np.where(dd.imag >= 0, np.where(dd.real >= 0, 0, 1), np.where(dd.real >= 0, 2, 3)).astype(np.uint8)
    return hard, dd, dict(t=ts, delta=delta)
