"""Synthetic IQ generators for tests and benchmarks (host side, numpy only).

The reference has no signal generator and no IQ file reader;
these are this project's own definitions:

* ``cu8`` wire format = rtl_sdr raw: interleaved unsigned bytes I,Q.
* ``cu8_to_c128`` = the pyrtlsdr conversion ``(u8 / 127.5) - 1`` per component
  (pyrtlsdr is an un-vendored dependency of the reference,
  ``requirements.txt:3``; this is the conversion its ``read_samples`` applies).
* ``noise_cu8`` = the bit-reproducible known-answer input of SURVEY.md §8(c):
  integer draws only, so it regenerates identically on any host.
* ``dqpsk_cu8`` = pi/4-DQPSK, RRC alpha=0.35, unit power, AWGN, optional
  carrier offset, quantised to cu8.  Uses libm transcendentals, so fixtures
  built from it are stored as bytes rather than regenerated from the seed.
"""

import numpy as np

SYMBOL_RATE = 18000.0
# phase step per dibit symbol, the mapping the reference documents
# (tetraear/signal/processor.py:106-110,146-150)
DPHI = np.array([np.pi / 4, 3 * np.pi / 4, -np.pi / 4, -3 * np.pi / 4])


def cu8_to_c128(u8):
    """Interleaved uint8 I,Q -> complex128, pyrtlsdr convention."""
    u8 = np.asarray(u8, dtype=np.uint8)
    return (u8[0::2].astype(np.float64) + 1j * u8[1::2].astype(np.float64)) / 127.5 - (1 + 1j)


def noise_cu8(n_samples, seed):
    """Uniform random bytes, 2*n_samples of them (I,Q interleaved)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=2 * n_samples, dtype=np.uint8)


def rrc_pulse(t, alpha=0.35):
    """Root-raised-cosine impulse response at times t (in symbol periods)."""
    t = np.asarray(t, dtype=np.float64)
    out = np.empty_like(t)
    eps = 1e-9
    z = np.abs(t) < eps
    s = np.abs(np.abs(t) - 1.0 / (4 * alpha)) < eps
    g = ~(z | s)
    out[z] = 1.0 - alpha + 4 * alpha / np.pi
    out[s] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                     + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
    tg = t[g]
    out[g] = (np.sin(np.pi * tg * (1 - alpha)) + 4 * alpha * tg * np.cos(np.pi * tg * (1 + alpha))) \
        / (np.pi * tg * (1 - (4 * alpha * tg) ** 2))
    return out


def dqpsk_baseband(n_samples, sample_rate, seed, symbol_rate=SYMBOL_RATE, alpha=0.35,
                   span=8, timing_offset=0.0):
    """Unit-power pi/4-DQPSK baseband at `sample_rate`, plus the dibit symbols sent.

    x(t) = sum_k a_k h(t/T - k), evaluated directly at the sample instants so
    any sample rate works (no integer samples-per-symbol requirement).
    """
    rng = np.random.default_rng(seed)
    T = sample_rate / symbol_rate  # samples per symbol (float)
    n_sym = int(np.ceil(n_samples / T)) + 2 * span + 2
    dibits = rng.integers(0, 4, size=n_sym, dtype=np.uint8)
    phase = np.cumsum(DPHI[dibits])
    a = np.exp(1j * phase)
    n = np.arange(n_samples, dtype=np.float64)
    ts = n / T + span + timing_offset  # position in symbol units; first `span` symbols are lead-in
    k0 = np.floor(ts).astype(np.int64)
    x = np.zeros(n_samples, dtype=np.complex128)
    for j in range(-span // 2, span // 2 + 1):
        k = k0 + j
        x += a[k] * rrc_pulse(ts - k, alpha)
    x /= np.sqrt(np.mean(np.abs(x) ** 2))
    return x, dibits


def quantise_cu8(x, scale=0.5):
    """complex -> interleaved uint8, round(127.5*(x*scale+1)) clipped to 0..255."""
    out = np.empty(2 * len(x), dtype=np.uint8)
    out[0::2] = np.clip(np.rint(127.5 * (x.real * scale + 1.0)), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(127.5 * (x.imag * scale + 1.0)), 0, 255).astype(np.uint8)
    return out


def dqpsk_cu8(n_samples, sample_rate=2.4e6, seed=1, esn0_db=20.0, carrier_offset=0.0,
              timing_offset=0.0):
    """One carrier: pi/4-DQPSK + AWGN (+ offset), quantised to cu8.  Returns (u8, dibits)."""
    x, dibits = dqpsk_baseband(n_samples, sample_rate, seed, timing_offset=timing_offset)
    rng = np.random.default_rng(seed + 1)
    T = sample_rate / SYMBOL_RATE
    # Es/N0 at symbol rate: noise variance per sample = T / 10^(EsN0/10) for unit signal power
    sigma2 = T / (10.0 ** (esn0_db / 10.0))
    w = rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)
    # cap the wideband noise so the cu8 range is not saturated: noise is white over fs,
    # only the in-channel part matters for Es/N0, keep it exact and scale the sum instead
    y = x + np.sqrt(sigma2 / 2.0) * w
    if carrier_offset:
        y = y * np.exp(2j * np.pi * carrier_offset * np.arange(n_samples) / sample_rate)
    peak = 4.0 * np.sqrt(1.0 + sigma2)
    return quantise_cu8(y, scale=1.0 / peak), dibits


def _dqpsk_cu8_job(job):
    n_samples, sample_rate, seed = job
    return dqpsk_cu8(n_samples, sample_rate, seed=seed)[0]


def dqpsk_cu8_streams(n_samples, sample_rate, seeds, workers=None):
    """`dqpsk_cu8` for many seeds -> uint8 [len(seeds)][2 * n_samples], every stream its own symbols and noise (SURVEY 8(d)
    C4: "1024 separate cu8 streams, seeds 1000 + i").  One stream costs ~0.3 s of numpy on arrays long enough for numpy to
    drop the interpreter lock, so more than a few of them are made by a pool of THREADS (no fork or spawn beside a HIP /
    RCCL context, no re-import of the caller's main module); the bytes are the ones
    `dqpsk_cu8(n_samples, sample_rate, seed)` returns, in the order of `seeds`."""
    import os
    seeds = [int(s) for s in seeds]
    out = np.empty((len(seeds), 2 * n_samples), dtype=np.uint8)
    jobs = [(int(n_samples), float(sample_rate), s) for s in seeds]
    if workers is None:
        workers = min(64, os.cpu_count() or 1)
    workers = max(1, min(int(workers), len(seeds) // 2))
    if workers == 1:
        for i, j in enumerate(jobs):
            out[i] = _dqpsk_cu8_job(j)
        return out
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as pool:
        for i, u8 in enumerate(pool.map(_dqpsk_cu8_job, jobs)):
            out[i] = u8
    return out


def multicarrier_cu8(n_samples, sample_rate, offsets_hz, seed0=100, esn0_db=20.0):
    """Sum of carriers at the given offsets in one wideband stream."""
    acc = np.zeros(n_samples, dtype=np.complex128)
    n = np.arange(n_samples, dtype=np.float64)
    all_dibits = []
    for i, f in enumerate(offsets_hz):
        x, dibits = dqpsk_baseband(n_samples, sample_rate, seed0 + i)
        acc += x * np.exp(2j * np.pi * f * n / sample_rate)
        all_dibits.append(dibits)
    rng = np.random.default_rng(seed0 - 1)
    T = sample_rate / SYMBOL_RATE
    sigma2 = T / (10.0 ** (esn0_db / 10.0))
    acc += np.sqrt(sigma2 / 2.0) * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    peak = 4.0 * np.sqrt(len(offsets_hz) + sigma2)
    return quantise_cu8(acc, scale=1.0 / peak), all_dibits


def grid_carriers(n_samples, sample_rate, channels, M, seed0=300, snr_db=25.0):
    """Sum of pi/4-DQPSK carriers on a channeliser grid: channel index k sits at k*fs/M (k >= M/2: negative
    frequencies), own symbol seed seed0 + i and timing offset 0.11 i each, white noise for `snr_db` Es/N0.
    Returns (complex128 stream, {channel: dibits sent})."""
    acc = np.zeros(n_samples, dtype=np.complex128)
    t = np.arange(n_samples, dtype=np.float64)
    dibs = {}
    for i, k in enumerate(channels):
        x, dib = dqpsk_baseband(n_samples, sample_rate, seed0 + i, timing_offset=0.11 * i)
        f = (k if k < M // 2 else k - M) * sample_rate / M
        acc += x * np.exp(2j * np.pi * f * t / sample_rate)
        dibs[k] = dib
    rng = np.random.default_rng(seed0 - 1)
    sigma2 = (sample_rate / SYMBOL_RATE) / 10 ** (snr_db / 10)
    acc += np.sqrt(sigma2 / 2) * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    return acc, dibs
