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


@pytest.mark.parametrize("pieces", [2, 4, 8])
def test_definition_in_pieces_joins_to_the_whole_chunks_symbols(pieces):
    """oracle/tetra_np.demod_gardner(segments=K) -- what the library does for batches that leave the device idle -- against
    the whole-chunk loop: the same symbols before the first seam, the same count (two loops' instants differ by whole symbol
    periods at their seam), the same decisions behind the seams (every loop has converged when it takes over), no error
    against what was sent; chunks too short for the warm-ups have no such form."""
    assert tetra_np.gardner_segments(4096, 72000.0) is None
    assert tetra_np.gardner_segments(32768, 72000.0, pieces=16) is None
    for fs, n, ppm in ((72000.0, 32768, 60.0), (72000.0, 33001, -120.0), (90000.0, 40960, 0.0)):
        geo = tetra_np.gardner_segments(n, fs, pieces=pieces)
        sps = fs / 18000.0
        assert (pieces - 1) * geo["seg_step"] + geo["n_v"] == n and geo["seam_in"] >= tetra_np.GARDNER_WARMUP_SYMBOLS * sps
        assert geo["seam_in"] < tetra_np.GARDNER_WARMUP_SYMBOLS * sps + pieces        # (no longer than the warm-up needs)
        x, dib = _gardner_case(n, fs, 77, 0.3, 120.0, 20.0, ppm)
        x = x.astype(np.complex128)
        h1, s1, i1 = tetra_np.demod_gardner(x, fs)
        h2, s2, i2 = tetra_np.demod_gardner(x, fs, segments=pieces)
        assert abs(len(s2) - len(s1)) <= 1
        k_seam = int(geo["seam_out"] / sps) - 40
        np.testing.assert_array_equal(i2["t"][:k_seam], i1["t"][:k_seam])        # (the second return value is derotated by the chunk's estimate)
        np.testing.assert_allclose(np.abs(s2[:k_seam - 1]), np.abs(s1[:k_seam - 1]), rtol=1e-12)
        m = min(len(h1), len(h2))
        assert np.mean(h1[:m] != h2[:m]) <= 1e-3
        assert np.max(np.abs(i2["t"][:m] - i1["t"][:m])) < 0.1 * sps          # one lattice of instants: no symbol lost or doubled
        best = min(int(np.sum(h2[300:m - 8] != dib[lag + 300:lag + m - 8])) for lag in range(40) if len(dib) - lag >= m)
        assert best == 0, (fs, n, best)


def _early_errors(hard, dib, lo=8, hi=400):
    """symbol errors against what was sent among symbols lo..hi of a chunk (lag found over the whole chunk)"""
    m = len(hard)
    lag = min(range(40), key=lambda g: int(np.sum(hard[600:m - 8] != dib[g + 600:g + m - 8])) if len(dib) - g >= m else 1 << 30)
    return int(np.sum(hard[lo:hi] != dib[lag + lo:lag + hi]))


def test_definition_with_a_feed_forward_start_acquires_at_once():
    """demod_gardner(ff_first=True) (the library's plan option gardner_ff_start): the chunk's first loop starts at the square-law
    estimate instead of at sample 1 + sps.  Over 48 carriers with random timing phases the plain loop loses symbols at the
    start of the chunks it begins half a symbol off the eye in (its detector's error vanishes there too); the feed-forward
    start does not -- and a thousand symbols in, both have made the same decisions."""
    fs, n = 72000.0, 12000
    plain = ff = 0
    worst_plain = 0
    for r in range(48):
        x, dib = _gardner_case(n, fs, 5200 + r, (r * 0.0213) % 1.0 - 0.5, float((r * 37) % 200 - 100), 20.0, float((r % 5) - 2) * 50.0)
        x = x.astype(np.complex128)
        h0, _, i0 = tetra_np.demod_gardner(x, fs)
        h1, _, i1 = tetra_np.demod_gardner(x, fs, ff_first=True)
        e0, e1 = _early_errors(h0, dib), _early_errors(h1, dib)
        plain, ff, worst_plain = plain + e0, ff + e1, max(worst_plain, e0)
        m = min(len(h0), len(h1))
        assert abs(len(h0) - len(h1)) <= 1
        d = 0 if np.array_equal(h0[1500:m - 4], h1[1500:m - 4]) else (1 if np.array_equal(h0[1501:m - 4], h1[1500:m - 5]) else -1)
        a, b = (h0[1500 + max(d, 0):m - 4], h1[1500 + max(-d, 0):m - 4])
        k = min(len(a), len(b))
        assert np.array_equal(a[:k], b[:k]), r
    print(f"symbol errors among symbols 8..400 of 48 chunks: plain start {plain} (worst chunk {worst_plain}), feed-forward start {ff}")
    assert ff == 0 and plain > 20


# ---- the device's Gardner receiver (TDM_MODE_TETRA_GARDNER) against the same definition ------------------------------
def _gardner_case(n, fs, seed, toff, coff, snr_db, rate_ppm=0.0):
    """a carrier whose symbol clock runs rate_ppm fast (the loop has to track a ramp, and the carriers of a wavefront drift
    apart)"""
    x, dib = synth.dqpsk_baseband(n, fs / (1.0 + rate_ppm * 1e-6), seed, timing_offset=toff)
    rng = np.random.default_rng(seed + 100)
    sps = fs / 18000.0
    sigma2 = sps / 10 ** (snr_db / 10)
    x = x + np.sqrt(sigma2 / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return (x * np.exp(2j * np.pi * coff * np.arange(n) / fs)).astype(np.complex64), dib


def _check_against_definition(x, fs, hard, soft, dib, skip=300, segments=1, ff_first=False):
    """segments: tdm_plan_info.gardner_segments of the plan that made `hard` (2: every chunk as two independently started
    loops joined at a seam -- the definition is then evaluated the same way)"""
    ref_hard, ref_dd, info = tetra_np.demod_gardner(x.astype(np.complex128), fs, segments=segments, ff_first=ff_first)
    # the loop runs in fp32 on the device (instants in fp64): symbol count within one of the definition's at the end of
    # the chunk, decisions equal wherever the definition's own derotated product is not within 0.1 rad of a quadrant
    # boundary (the bar tools/sweep_gardner.py uses) -- and never more than 1e-3 of them
    assert abs(len(soft) - len(info["t"])) <= 1, (len(soft), len(info["t"]))
    m = min(len(hard), len(ref_hard))
    assert m > 0.9 * len(x) / (fs / 18000.0) - 20
    diff = np.flatnonzero(hard[:m] != ref_hard[:m])
    assert len(diff) <= 1e-3 * m, len(diff) / m
    ang = np.angle(ref_dd[diff])
    assert np.all(np.abs((ang + np.pi / 4) % (np.pi / 2) - np.pi / 4) < 0.1), (diff, ang)
    best = min((int(np.sum(hard[skip:m - 8] != dib[lag + skip:lag + m - 8])) for lag in range(40) if len(dib) - lag >= m), default=-1)
    return best


@pytest.mark.gpu
def test_gpu_gardner_receiver_matches_definition_and_transmitted():
    """Gardner TED + PI loop + Farrow on the device, four lanes per carrier: against oracle/tetra_np.demod_gardner (decisions)
    and against the transmitted dibits (error-free after the loop's acquisition at 20 dB), at 3..8 samples per symbol,
    with timing and carrier offsets and a symbol clock 200 ppm off"""
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    for fs, n, ppm in ((72000.0, 16384, 0.0), (72000.0, 32768, 200.0), (80000.0, 12000, -150.0), (54000.0, 6000, 0.0), (144000.0, 20000, 100.0)):
        rows = 3
        sig = [_gardner_case(n, fs, 40 + 7 * r, 0.13 * r - 0.2, (-120.0, 0.0, 90.0)[r], 20.0, ppm) for r in range(rows)]
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        seg = bd.info.gardner_segments
        hards, softs, timing, margin = bd.process(np.concatenate([s[0] for s in sig]))
        bd.close()
        for r in range(rows):
            errs = _check_against_definition(sig[r][0], fs, hards[r], softs[r], sig[r][1], segments=seg)
            assert errs == 0, (fs, n, r, errs)
            assert 0.0 < margin[r] < 0.8


@pytest.mark.gpu
def test_gpu_gardner_receiver_many_carriers_share_a_wavefront():
    """130 carriers (eight full wavefronts of 16 and a partial one) with different symbol-clock offsets: the carriers of a
    wavefront drift apart by several samples over the chunk while sharing one ring of matched-filter samples; every one
    equals the definition"""
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    fs, n, rows = 72000.0, 24576, 130
    sig = [_gardner_case(n, fs, 500 + r, ((r * 37) % 100) / 100.0 - 0.5, float((r * 53) % 240 - 120), 18.0, float((r % 9) - 4) * 100.0)
           for r in range(rows)]
    bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
    seg = bd.info.gardner_segments
    hards, softs, timing, margin = bd.process(np.concatenate([s[0] for s in sig]))
    bd.close()
    for r in range(0, rows, 3):
        _check_against_definition(sig[r][0], fs, hards[r], softs[r], sig[r][1], segments=seg)
    # every carrier (also the ones not compared with the slow fp64 loop): error-free against what was sent, after acquisition
    for r in range(rows):
        m = len(hards[r])
        dib = sig[r][1]
        assert min(int(np.sum(hards[r][300:m - 8] != dib[lag + 300:lag + m - 8])) for lag in range(40) if len(dib) - lag >= m) == 0, r


@pytest.mark.gpu
def test_gpu_gardner_carriers_of_a_wavefront_more_than_three_chunks_apart():
    """Symbol clocks on time, 1 % slow and 0.3 % fast (more symbols than the nominal count: the row has room for a clock
    2 % fast) in ONE wavefront: over 24 576 samples the carriers drift 245 samples apart,
    more than the three chunks of the shared ring -- the fast ones wait for the ring to move on (turns taken lane by lane
    instead of the straight-line run).  Every carrier equals the definition (a loop of this bandwidth slips symbols while it pulls
    in a 1 % offset -- the definition's does too --, so only the carriers on time are also held against what was sent)."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, debug_option
    from tetraear_amd.batch import BatchDemodulator
    fs, n, rows = 72000.0, 24576, 20
    sig = [_gardner_case(n, fs, 900 + r, 0.1 * (r % 5) - 0.2, float((r * 31) % 200 - 100), 25.0, (3000.0 if r % 4 == 1 else 0.0) if r % 2 else -10000.0)
           for r in range(rows)]
    # whole chunks (the drift of a whole chunk inside one wavefront), then the chunks in the pieces the plan picks (each
    # piece's loop pulls the clock offset in anew: other slips, the same definition evaluated the same way)
    for allow in (0, 1):
        with debug_option("gardner_segments", allow):
            bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
            seg = bd.info.gardner_segments
            assert (seg == 1) if allow == 0 else (seg > 1)
            hards, softs, timing, margin = bd.process(np.concatenate([s[0] for s in sig]))
            bd.close()
        counts = [len(s) for s in softs]
        if seg == 1:
            assert max(counts) - min(counts) >= 55, counts      # (1 % of 6144 symbols: the groups really are that far apart)
        assert max(counts) > n / (fs / 18000.0) + 10, counts    # (the fast carriers: more symbols than the nominal count)
        for r in range(rows):
            errs = _check_against_definition(sig[r][0], fs, hards[r], softs[r], sig[r][1], skip=600, segments=seg)
            assert errs == 0 or r % 4 != 3, (seg, r, errs)           # (r % 4 == 3: the carriers on time)


@pytest.mark.gpu
def test_gpu_gardner_short_rows_silence_and_capacity():
    """rows as short as a plan takes (64 samples: one ring chunk, one block of turns), a silent carrier beside live ones (zero error, nominal
    period: at 4 samples per symbol its last instant is n - 3 exactly, the closed end of the chunk), odd lengths and
    pitches: symbol counts and decisions against the definition"""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, check, ptr
    from tetraear_amd.batch import BatchDemodulator
    fs = 72000.0
    for n in (64, 65, 67, 131, 257, 1000, 4096):
        rows, pitch = 5, n + 3
        xs = [_gardner_case(n, fs, 70 + r, 0.2 * r - 0.3, 0.0, 30.0)[0] for r in range(rows)]
        xs[2] = np.zeros(n, np.complex64)
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        ms = bd.info.max_soft
        buf = np.full((rows, pitch), 7.0 + 7.0j, dtype=np.complex64)
        for r in range(rows):
            buf[r, :n] = xs[r]
        hard = np.full((rows, ms), 9, np.uint8)
        soft = np.zeros((rows, ms), np.complex64)
        ns = np.zeros(rows, np.int32)
        check(bd.lib.tdm_process(bd.handle, ptr(buf), pitch, None, None, ptr(hard), ptr(soft), ptr(ns), None, None))
        bd.close()
        for r in range(rows):
            ref_hard, _, info = tetra_np.demod_gardner(xs[r].astype(np.complex128), fs)
            if r == 2:
                assert ns[r] == len(info["t"]), (n, ns[r], len(info["t"]))      # exact arithmetic on zeros: the same count
                assert not np.any(soft[r, :ns[r]])
                continue
            assert abs(int(ns[r]) - len(info["t"])) <= 1, (n, r, ns[r], len(info["t"]))
            m = min(max(int(ns[r]) - 1, 0), len(ref_hard))
            assert int(np.sum(hard[r, :m] != ref_hard[:m])) <= (1 if m > 200 else 0), (n, r)


@pytest.mark.gpu
def test_gpu_gardner_fused_kernel_agrees_with_the_three_launches():
    """the default path (matched filter by producer wavefronts inside the loop's workgroup, filter output in LDS only) against
    the three-launch path (k_tetra_mf -> HBM -> loop; tdm_debug_set gardner_fused 0) on the same batch: the same symbol counts and
    decisions, soft symbols within fp32 rounding -- at every tap count a plan can have (3..8 samples per symbol), with rows
    that are no multiple of a wavefront's sixteen carriers and a pitch that is not the row length"""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, check, debug_option, ptr
    from tetraear_amd.batch import BatchDemodulator
    for fs, n in ((54000.0, 5000), (72000.0, 8192), (75000.0, 6001), (90000.0, 7000), (108000.0, 9000), (126000.0, 8000), (144000.0, 12000)):
        rows, pitch = 21, n + 5
        xs = [_gardner_case(n, fs, 300 + r, 0.04 * r - 0.4, float(r * 9 - 90), 22.0, float((r % 5) - 2) * 150.0)[0] for r in range(rows)]
        buf = np.full((rows, pitch), 3.0 - 2.0j, dtype=np.complex64)
        for r in range(rows):
            buf[r, :n] = xs[r]
        outs = []
        for fused in (1, 0):
            # (whole chunks on both sides: the comparison is between the kernels, and only the fused one walks chunks in pieces)
            with debug_option("gardner_fused", fused), debug_option("gardner_segments", 0):
                bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
                ms = bd.info.max_soft
                hard = np.zeros((rows, ms), np.uint8)
                soft = np.zeros((rows, ms), np.complex64)
                ns = np.zeros(rows, np.int32)
                tm = np.zeros(rows, np.int32)
                check(bd.lib.tdm_process(bd.handle, ptr(buf), pitch, None, None, ptr(hard), ptr(soft), ptr(ns), ptr(tm), None))
                bd.close()
            outs.append((hard, soft, ns, tm))
        (h1, s1, n1, t1), (h0, s0, n0, t0) = outs
        assert np.array_equal(n1, n0), (fs, n1, n0)
        assert np.array_equal(t1, t0)
        for r in range(rows):
            k = int(n1[r])
            assert k > 0.9 * n / (fs / 18000.0) - 10
            assert np.array_equal(h1[r, :k - 1], h0[r, :k - 1]), (fs, r)
            scale = float(np.max(np.abs(s0[r, :k])))
            assert float(np.max(np.abs(s1[r, :k] - s0[r, :k]))) <= 2e-6 * scale, (fs, r)


@pytest.mark.gpu
def test_gpu_gardner_more_carriers_than_one_round_of_workgroups():
    """4200 carriers = 263 workgroups of the fused kernel on 256 compute units: at 4 samples per symbol two workgroups share a
    compute unit, at 8 one fits and the call takes the three launches instead; either way every carrier equals what the
    three launches give"""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, debug_option
    from tetraear_amd.batch import BatchDemodulator
    for fs in (72000.0, 144000.0):
        n, rows, distinct = 3000, 4200, 24
        base = [_gardner_case(n, fs, 1200 + r, 0.03 * r - 0.35, float(r * 7 - 80), 22.0, float((r % 7) - 3) * 100.0)[0] for r in range(distinct)]
        iq = np.concatenate([base[r % distinct] for r in range(rows)])
        outs = []
        for fused in (1, 0):
            with debug_option("gardner_fused", fused):
                bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
                outs.append(bd.process(iq))
                bd.close()
        (h1, s1, t1, m1), (h0, s0, t0, m0) = outs
        assert np.array_equal(t1, t0)
        for r in range(rows):
            assert len(s1[r]) == len(s0[r]) and len(h1[r]) > 0.9 * n / (fs / 18000.0) - 10, (fs, r)
            assert np.array_equal(h1[r], h0[r]), (fs, r)
            assert np.array_equal(h1[r], h1[r % distinct]), (fs, r)      # (and its prototype row)


@pytest.mark.gpu
def test_gpu_gardner_pieces_per_carrier_against_whole_chunks():
    """tdm_plan_info.gardner_segments = 2, 4 or 8 (batches that would leave most of the device idle, chunks long enough):
    every carrier's chunk as that many independently started loops joined at seams.  Against the same definition evaluated
    in pieces (decisions, counts), against what was sent (no error behind the seams either), and against the device's own
    whole-chunk path (tdm_debug_set gardner_segments 0): the same symbol count, decisions equal up to 1e-3 of them, soft
    symbols equal bit for bit before the first seam and within 15 % right behind a seam (the loop that takes over is 512
    symbols into its run there: a few per cent of a symbol of timing error left, shrinking with the loop's time constant)
    -- at 4, 5 and 8 samples per symbol (65 taps: one workgroup per compute unit), with rows that are and are not a
    multiple of the loop wavefront's sixteen carriers, the number of pieces the plan picks and the smaller ones
    (tdm_debug_set gardner_segments K: at most K).  Whole chunks stay where the rule says so: chunks too short for the
    warm-ups, batches that fill the device by themselves."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, debug_option
    from tetraear_amd.batch import BatchDemodulator
    # the default follows from the chunk alone (length, rate, taps): too short for the warm-ups -> whole chunks; long enough ->
    # 8 pieces whatever the batch (8200 and 4096 carriers run in several rounds of workgroups); 65 taps (8 samples per
    # symbol: one workgroup per compute unit, the fused kernel serves one round only) -> whole chunks.  Fitted to the
    # batch (-1, the round-5 rule): 2 pieces at the bench leg's 4096 carriers, whole chunks once the batch fills the device.
    for fs, n, rows, want, fitted in ((72000.0, 4096, 8, 1, 1), (72000.0, 32768, 8200, 8, 1), (72000.0, 32768, 4096, 8, 2),
                                     (72000.0, 32768, 16, 8, 8), (144000.0, 65536, 16, 1, 8)):
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        assert bd.info.gardner_segments == want, (fs, n, rows, bd.info.gardner_segments)
        bd.set_gardner_segments(-1)
        assert bd.info.gardner_segments == fitted, (fs, n, rows, bd.info.gardner_segments)
        bd.close()
    for fs, n, rows, picks, rule in ((72000.0, 32768, 32, 8, 1), (72000.0, 30001, 21, 8, 1), (90000.0, 40960, 16, 8, 1), (144000.0, 65536, 16, 8, -1)):
        sig = [_gardner_case(n, fs, 1500 + r, 0.07 * r - 0.4, float((r * 29) % 200 - 100), 20.0, float((r % 5) - 2) * 60.0) for r in range(rows)]
        iq = np.concatenate([s[0] for s in sig])
        with debug_option("gardner_segments", 0):
            bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
            assert bd.info.gardner_segments == 1
            h1, s1, t1, m1 = bd.process(iq)
            bd.close()
        seen = set()
        for at_most in ((rule, 2, 4) if rule == 1 else (rule,)):       # 1: the default; -1: fitted to the batch (65 taps: the only way to pieces)
            with debug_option("gardner_segments", at_most):
                bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
                K = bd.info.gardner_segments
                assert K == (picks if at_most in (1, -1) else min(picks, at_most)), (fs, n, rows, at_most, K)
                if K in seen:
                    bd.close()
                    continue
                seen.add(K)
                h2, s2, t2, m2 = bd.process(iq)
                bd.close()
            geo = tetra_np.gardner_segments(n, fs, pieces=K)
            k_seam = int(geo["seam_out"] / (fs / 18000.0))
            worst = 0.0
            for r in range(rows):
                assert abs(len(s2[r]) - len(s1[r])) <= 1, (fs, K, r, len(s2[r]), len(s1[r]))
                m = min(len(h1[r]), len(h2[r]))
                assert np.mean(h1[r][:m] != h2[r][:m]) <= 1e-3, (fs, K, r)
                np.testing.assert_array_equal(s2[r][:k_seam - 40], s1[r][:k_seam - 40])
                scale = float(np.max(np.abs(s1[r])))
                dev = float(np.max(np.abs(s2[r][:m] - s1[r][:m]))) / scale
                worst = max(worst, dev)
                assert dev <= 0.05, (fs, K, r, dev)
                assert float(np.max(np.abs(s2[r][m - 300:m] - s1[r][m - 300:m]))) <= 0.02 * scale, (fs, K, r)
                # (the timing phase at the chunk's middle symbol, in 1/1000 symbol: that very symbol of the same loop for two
                #  pieces; for more, a neighbour of it in a loop that took over a few hundred symbols earlier: within 1 % of a symbol)
                assert (t2[r] == t1[r]) if K == 2 else (min((t2[r] - t1[r]) % 1000, (t1[r] - t2[r]) % 1000) <= 10), (fs, K, r, t2[r], t1[r])
                if r % 4 == 0:
                    errs = _check_against_definition(sig[r][0], fs, h2[r], s2[r], sig[r][1], segments=K)
                    assert errs == 0, (fs, K, r, errs)
            print(f"fs {fs:.0f} n {n} rows {rows} pieces {K}: largest soft-symbol deviation from the whole-chunk path {worst:.4f} of the largest symbol")


@pytest.mark.gpu
def test_gpu_gardner_segments_as_a_plan_option():
    """tdm_plan_option "gardner_segments" on a live plan: whole chunks, the rule's choice, a cap -- each time the outputs of a
    plan made that way; options of the wrong kind are refused."""
    from tetraear_amd._lib import MODE_TETRA, MODE_TETRA_GARDNER, TetraHipError, debug_option
    from tetraear_amd.batch import BatchDemodulator
    fs, n, rows = 72000.0, 32768, 16
    sig = [_gardner_case(n, fs, 3100 + r, 0.05 * r - 0.3, float(r * 11 - 80), 20.0, float((r % 3) - 1) * 80.0) for r in range(rows)]
    iq = np.concatenate([s[0] for s in sig])
    fresh = {}
    for allow in (0, 1, 2):
        with debug_option("gardner_segments", allow):
            bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
            fresh[allow] = (bd.info.gardner_segments, bd.process(iq))
            bd.close()
    assert [fresh[a][0] for a in (0, 1, 2)] == [1, 8, 2]
    bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
    for allow in (2, 0, 1, 0):
        bd.set_gardner_segments(allow)
        assert bd.info.gardner_segments == fresh[allow][0]
        h, s, t, m = bd.process(iq)
        for r in range(rows):
            assert np.array_equal(h[r], fresh[allow][1][0][r]) and np.array_equal(s[r], fresh[allow][1][1][r]), (allow, r)
        assert np.array_equal(t, fresh[allow][1][2])
    with pytest.raises(TetraHipError):
        bd.set_gardner_segments(9)
    with pytest.raises(TetraHipError):
        bd.set_gardner_segments(-2)
    bd.close()
    bf = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
    with pytest.raises(TetraHipError):
        bf.set_gardner_segments(0)
    bf.close()


@pytest.mark.gpu
def test_gpu_gardner_pieces_with_a_silent_carrier_and_a_fast_clock():
    """In pieces: a carrier of zeros (no error signal, no timing estimate to start from: every loop runs at the nominal
    period from the nominal start, and the seams still join to the nominal count) and one whose symbol clock runs 0.3 % fast
    (more symbols than nominal in every piece) beside ordinary ones -- counts and decisions as the definition evaluated
    the same way."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    fs, n, rows = 72000.0, 32768, 16
    sig = [_gardner_case(n, fs, 4100 + r, 0.06 * r - 0.4, float(r * 13 - 90), 22.0, 3000.0 if r == 5 else 0.0) for r in range(rows)]
    sig[9] = (np.zeros(n, np.complex64), sig[9][1])
    bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
    K = bd.info.gardner_segments
    assert K == 8
    hards, softs, timing, margin = bd.process(np.concatenate([s[0] for s in sig]))
    bd.close()
    nominal = n / (fs / 18000.0)
    assert abs(len(softs[9]) - nominal) <= 3 and not np.any(softs[9]) and not np.any(hards[9])
    assert len(softs[5]) > nominal + 8
    for r in range(rows):
        ref_hard, _, info = tetra_np.demod_gardner(sig[r][0].astype(np.complex128), fs, segments=K)
        assert abs(len(softs[r]) - len(info["t"])) <= 1, (r, len(softs[r]), len(info["t"]))
        if r not in (5, 9):
            assert _check_against_definition(sig[r][0], fs, hards[r], softs[r], sig[r][1], segments=K) == 0, r
        elif r == 5:      # (a 0.3 % clock offset: a loop may slip while it pulls in, in the definition as on the device)
            m = min(len(hards[r]), len(ref_hard))
            assert np.mean(hards[r][:m] != ref_hard[:m]) <= 0.02, r


@pytest.mark.gpu
def test_gpu_gardner_feed_forward_start_option():
    """tdm_plan_option "gardner_ff_start": every chunk's first loop starts at the feed-forward estimate -- as whole chunks and
    in pieces the outputs are the definition's evaluated the same way (demod_gardner(ff_first=True)), chunks that the plain
    start begins half a symbol off the eye lose no symbols at their start any more, and a plan the fused kernel does not
    serve refuses the option."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER, TetraHipError, debug_option
    from tetraear_amd.batch import BatchDemodulator
    fs, n, rows = 72000.0, 12000, 48
    sig = [_gardner_case(n, fs, 5200 + r, (r * 0.0213) % 1.0 - 0.5, float((r * 37) % 200 - 100), 20.0, float((r % 5) - 2) * 50.0) for r in range(rows)]
    iq = np.concatenate([s[0] for s in sig])
    early = {}
    for allow in (0, 1):
        for ff in (False, True):
            with debug_option("gardner_segments", allow):
                bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
            K = bd.info.gardner_segments
            assert (K == 1) if allow == 0 else (K > 1)
            if ff:
                bd.set_gardner_ff_start(True)
            hards, softs, timing, margin = bd.process(iq)
            bd.close()
            early[(K, ff)] = sum(_early_errors(hards[r], sig[r][1]) for r in range(rows))
            for r in range(0, rows, 3):
                ref_hard, ref_dd, info = tetra_np.demod_gardner(sig[r][0].astype(np.complex128), fs, segments=K, ff_first=ff)
                assert abs(len(softs[r]) - len(info["t"])) <= 1, (K, ff, r)
                m = min(len(hards[r]), len(ref_hard))
                diff = np.flatnonzero(hards[r][:m] != ref_hard[:m])
                # (a loop that hangs half a symbol off the eye decides on noise: only the feed-forward start is held to the tight bar)
                assert len(diff) <= (1e-3 if ff else 2e-2) * m, (K, ff, r, len(diff))
    print("symbol errors among symbols 8..400 of 48 chunks, (pieces, feed-forward start):", early)
    whole = [k for k in early if k[0] == 1]
    assert early[(1, True)] == 0 and early[(1, False)] > 20, early
    assert all(v == 0 for k, v in early.items() if k[1]), early
    with debug_option("gardner_fused", 0):
        bd = BatchDemodulator(fs, n, 4, "cf32", mode=MODE_TETRA_GARDNER)
        with pytest.raises(TetraHipError):
            bd.set_gardner_ff_start(True)
        bd.close()


@pytest.mark.gpu
def test_gpu_gardner_result_does_not_depend_on_the_batch():
    """Round-5 review: "a receiver's result must not depend on who else is in the launch".  The same 16 carriers demodulated
    inside plans of 16, 1024 and 4096 carriers (the rest of each batch other signals): the default number of pieces is the
    same in all three (it follows from the chunk alone) and the 16 carriers' hard decisions, soft symbols, counts and timing
    come out bit for bit the same -- also where they sit in the batch (first rows, last rows).  With the pieces fitted to the
    batch (plan option -1) the number differs between those plans, which is what the default no longer does."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    fs, n, keep = 72000.0, 32768, 16
    sig = [_gardner_case(n, fs, 7100 + r, 0.06 * r - 0.45, float((r * 23) % 200 - 100), 20.0, float((r % 5) - 2) * 70.0)[0] for r in range(keep)]
    filler = [_gardner_case(n, fs, 7300 + r, 0.11 * r - 0.3, float(r * 17 - 60), 18.0, 0.0)[0] for r in range(8)]
    ref, pieces, fitted = None, set(), set()
    for rows, at in ((16, 0), (1024, 0), (1024, 1008), (4096, 0), (4096, 4080)):
        iq = np.empty((rows, n), np.complex64)
        for r in range(rows):
            iq[r] = filler[r % len(filler)]
        iq[at:at + keep] = sig
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        pieces.add(int(bd.info.gardner_segments))
        hards, softs, timing, margin = bd.process(iq.reshape(-1))
        bd.set_gardner_segments(-1)
        fitted.add(int(bd.info.gardner_segments))
        bd.close()
        got = ([hards[at + r] for r in range(keep)], [softs[at + r] for r in range(keep)], np.array(timing[at:at + keep]), np.array(margin[at:at + keep]))
        if ref is None:
            ref = got
            assert all(len(h) > 8000 for h in got[0])
            continue
        for r in range(keep):
            assert np.array_equal(got[0][r], ref[0][r]), (rows, at, r)
            assert np.array_equal(got[1][r], ref[1][r]), (rows, at, r)
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), (rows, at)
    assert pieces == {8}, pieces
    assert fitted == {8, 2}, fitted
