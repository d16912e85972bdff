"""Randomised sweep of TDM_MODE_TETRA_GARDNER on the GPU: random chunk lengths, sample rates (3..8 samples/symbol), row
strides, timing / carrier / symbol-clock offsets at 15..25 dB; decisions against the fp64 definition's loop
(oracle/tetra_np.demod_gardner: decisions may differ where the definition's own derotated product lies within 0.1 rad of a quadrant boundary -- an fp32 loop against an fp64 one at 15 dB --, at most 2e-3 of them; count within one
-- evaluated in as many pieces per chunk as the plan runs, tdm_plan_info.gardner_segments); also the stand-alone RRC filter
against the definition."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import tetra_np
from tetraear_amd import synth
from tetraear_amd._lib import MODE_TETRA_GARDNER, check, debug_option, ptr
from tetraear_amd.batch import BatchDemodulator
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time(); bad = 0; cnt = 0; worst = 0.0; halves = {}
sent = {"pieces": [0, 0], "whole": [0, 0]}
n_ff = 0
while time.time() - t0 < budget:
    fs = float(rng.choice([54000.0, 72000.0, 75000.0, 80000.0, 90000.0, 108000.0, 144000.0]))
    n = int(rng.integers(3000, 20000)) if rng.random() < 0.5 else int(rng.integers(20000, 70000))   # (the longer ones run in pieces)
    rows = int(rng.integers(1, 4)) if rng.random() < 0.7 else int(rng.integers(15, 35))
    pitch = n + int(rng.integers(0, 9))
    xs, dibs = [], []
    for r in range(rows):
        ppm = float(rng.uniform(-300, 300))
        snr = float(rng.uniform(15, 25))
        x, d = synth.dqpsk_baseband(n, fs / (1 + ppm * 1e-6), int(rng.integers(1 << 30)), timing_offset=float(rng.uniform(-0.5, 0.5)))
        sps = fs / 18000.0
        x = x + np.sqrt(sps / 10 ** (snr / 10) / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        x = x * np.exp(2j * np.pi * float(rng.uniform(-100, 100)) * np.arange(n) / fs)
        xs.append(x.astype(np.complex64)); dibs.append(d)
    bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
    ff = bool(rng.random() < 0.5)      # plan option gardner_ff_start on half of the plans
    if ff:
        bd.set_gardner_ff_start(True)
    n_ff += rows if ff else 0
    buf = np.full((rows, pitch), 9.0, dtype=np.complex64)
    for r in range(rows): buf[r, :n] = xs[r]
    ms = bd.info.max_soft
    hard = np.zeros((rows, ms), np.uint8); soft = np.zeros((rows, ms), np.complex64); ns = np.zeros(rows, np.int32)
    check(bd.lib.tdm_process(bd.handle, ptr(buf), pitch, None, None, ptr(hard), ptr(soft), ptr(ns), None, None))
    y = bd.rrc_filter(np.stack(xs))
    pieces = bd.info.gardner_segments
    if pieces > 1:      # the same batch as whole chunks: symbol errors against what was sent, both ways
        with debug_option("gardner_segments", 0):
            bw = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
            if ff:
                bw.set_gardner_ff_start(True)
            hw = np.zeros((rows, ms), np.uint8); sw = np.zeros((rows, ms), np.complex64); nw = np.zeros(rows, np.int32)
            check(bw.lib.tdm_process(bw.handle, ptr(buf), pitch, None, None, ptr(hw), ptr(sw), ptr(nw), None, None))
            bw.close()
        for r in range(rows):
            for tag, hh, nn_ in (("pieces", hard, ns), ("whole", hw, nw)):
                h = hh[r, :max(nn_[r] - 1, 0)]
                m = len(h)
                e = min((int(np.sum(h[300:m - 8] != dibs[r][lag + 300:lag + m - 8])) for lag in range(40) if len(dibs[r]) - lag >= m), default=0)
                sent[tag][0] += e; sent[tag][1] += max(m - 308, 0)
    for r in range(rows):
        cnt += 1
        h = hard[r, :max(ns[r] - 1, 0)]
        rh, rdd, info = tetra_np.demod_gardner(xs[r].astype(np.complex128), fs, segments=bd.info.gardner_segments, ff_first=ff)
        halves[bd.info.gardner_segments] = halves.get(bd.info.gardner_segments, 0) + 1
        m = min(len(h), len(rh))
        diff = np.flatnonzero(h[:m] != rh[:m])
        frac = len(diff) / m if m else 1.0
        # distance of the definition's derotated products from the nearest quadrant boundary, at the differing decisions
        ang = np.angle(rdd[diff]) if len(diff) else np.zeros(0)
        marginal = bool(np.all(np.abs((ang + np.pi / 4) % (np.pi / 2) - np.pi / 4) < 0.1))
        worst = max(worst, frac)
        ref = tetra_np.matched_filter(xs[r].astype(np.complex128), tetra_np.rrc_taps(fs / 18000.0))
        mf_err = float(np.max(np.abs(y[r] - ref)) / np.max(np.abs(ref)))
        ok = abs(int(ns[r]) - len(info["t"])) <= 1 and marginal and frac <= 2e-3 and mf_err < 2e-6
        if not ok:
            bad += 1; print("MISMATCH", fs, n, rows, pitch, ns[r], len(info["t"]), frac, mf_err)
    bd.close()
print("symbol errors against what was sent, carriers run in pieces: in pieces %d of %d, as whole chunks %d of %d" % (*sent["pieces"], *sent["whole"]))
print(f"{n_ff} of the carriers with the plan option gardner_ff_start")
print(f"{cnt} carriers (pieces per chunk: {dict(sorted(halves.items()))}), {bad} mismatches, worst fraction of differing decisions {worst:.2e}")
