"""CPU: the feed-forward timing recovery the device runs (oracle/tetra_np.demod, which the HIP kernels match decision
for decision, tests/test_tetra_mode.py) against the textbook loop BASELINE.json's north_star names -- Gardner
timing-error detector + PI loop + Farrow interpolator (oracle/tetra_np.demod_gardner, fp64, sequential).

Sweep: Es/N0 5..25 dB x timing offset 0 / 0.25 / 0.5 symbol x carrier offset -200 / 0 / +200 Hz, two noise/data seeds
per point, 4 samples per symbol, 16384-sample chunks; the first 300 symbols (the loop's acquisition) are excluded for
BOTH receivers.  At every point the feed-forward receiver must (a) make no more symbol errors than the loop, up to
three standard deviations of the error count, and (b) have no more RMS timing error.  The loop runs at a noise
bandwidth of 1 % of the symbol rate; the feed-forward estimate averages 5 sub-blocks of 256 samples."""
import numpy as np
import pytest

from oracle import tetra_np
from tetraear_amd import synth

FS, N, SKIP = 72000.0, 16384, 300


def _signal(seed, toff, coff, snr_db):
    x, dib = synth.dqpsk_baseband(N, FS, seed, timing_offset=toff)
    rng = np.random.default_rng(seed + 100)
    sps = FS / 18000.0
    sigma2 = sps / 10 ** (snr_db / 10)
    x = x + np.sqrt(sigma2 / 2) * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    return x * np.exp(2j * np.pi * coff * np.arange(N) / FS), dib


def _errors(hard, dib):
    best = (1 << 30, 1)
    for lag in range(40):
        m = min(len(hard), len(dib) - lag)
        if m < SKIP + 100:
            continue
        e = int(np.sum(hard[SKIP:m - 8] != dib[lag + SKIP:lag + m - 8]))
        if e < best[0]:
            best = (e, m - 8 - SKIP)
    return best


def _jitter(t, toff):
    sps = FS / 18000.0
    ph = (t / sps + toff + 0.5) % 1.0 - 0.5          # true symbol instants sit at (k - toff) * sps
    return float(np.sqrt(np.mean(ph[SKIP:-8] ** 2)))


@pytest.mark.parametrize("snr_db", [5, 10, 15, 20, 25])
def test_feed_forward_matches_or_beats_gardner(snr_db):
    rows = []
    for toff in (0.0, 0.25, 0.5):
        for coff in (-200.0, 0.0, 200.0):
            ef = eg = nf = ng = 0
            jf, jg = [], []
            for seed in (11, 12):
                x, dib = _signal(seed, toff, coff, snr_db)
                hf, _, inf = tetra_np.demod(x, FS)
                hg, _, ing = tetra_np.demod_gardner(x, FS)
                a, b = _errors(hf, dib)
                ef, nf = ef + a, nf + b
                a, b = _errors(hg, dib)
                eg, ng = eg + a, ng + b
                jf.append(_jitter(inf["t"], toff))
                jg.append(_jitter(ing["t"], toff))
            ser_f, ser_g = ef / nf, eg / ng
            sigma = np.sqrt(max(ser_g * (1 - ser_g), 1.0 / ng) / ng)
            rows.append((toff, coff, ser_f, ser_g, float(np.mean(jf)), float(np.mean(jg))))
            assert ser_f <= ser_g + 3 * sigma + 1e-12, rows[-1]
            assert np.mean(jf) <= np.mean(jg), rows[-1]
    print(f"Es/N0 {snr_db} dB: (timing offset, CFO Hz, SER feed-forward, SER Gardner, RMS timing error ff, Gardner [symbols])")
    for r in rows:
        print("   %.2f %6.0f  %.5f %.5f  %.4f %.4f" % r)
