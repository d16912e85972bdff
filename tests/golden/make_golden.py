#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (`/root/reference`, syrex1013/TetraEar v2.2) is imported read-only;
none of its source is copied.  The reference's own tests pin no numeric result on
this path, so these vectors ARE the parity anchor: outputs of
`tetraear.signal.processor.SignalProcessor` (processor.py:18-273) with
numpy/scipy versions recorded in the manifest.

Writes (all under tests/golden/):
  manifest.json          case metadata (+ sha256 of every hard-symbol output)
  inputs.npz             cu8 input bytes for the cases that are not seed-reproducible
  process.npz            <case>__hard (uint8), <case>__soft (complex128)
  stages.npz             per-method goldens (filter_signal, frequency_shift,
                         extract_symbols, demodulate_dqpsk, resample, decimate)
  design.npz             scipy filter-design tables (cheby1 SOS, butter b/a, zi)
"""
import hashlib
import json
import os
import sys

import numpy as np
import scipy
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from tetraear.signal.processor import SignalProcessor  # noqa: E402  (the reference)
from tetraear_amd import synth  # noqa: E402  (our generator)

RATES = [0.225e6, 0.9e6, 1.024e6, 1.536e6, 1.8e6, 1.92e6, 2.048e6, 2.4e6, 2.56e6, 2.88e6, 3.2e6, 10e6]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    manifest = {"numpy": np.__version__, "scipy": scipy.__version__,
                "reference": "syrex1013/TetraEar v2.2 tetraear/signal/processor.py",
                "cases": [], "stage_cases": []}
    inputs, proc, stages, design = {}, {}, {}, {}

    # ------------------------------------------------------------------ process() cases
    cases = []
    # KAT of SURVEY §8(c)
    cases.append(dict(name="kat_2400_f0", fs=2.4e6, foff=0.0, kind="noise", seed=20260929, n=131072))
    cases.append(dict(name="kat_2400_f1171", fs=2.4e6, foff=1171.875, kind="noise", seed=20260929, n=131072))
    # script chunk size
    cases.append(dict(name="noise_2400_256k", fs=2.4e6, foff=-3515.625, kind="noise", seed=7, n=262144))
    # every supported rate, short noise input (exercises q, sps, phase step)
    for i, fs in enumerate(RATES):
        cases.append(dict(name=f"rate_{int(fs/1e3)}k", fs=fs, foff=[0.0, 1171.875, -3515.625][i % 3],
                          kind="noise", seed=1000 + i, n=32768))
    # a rate off the RTL list (GUI slider is continuous 1.8-10 MHz)
    cases.append(dict(name="rate_2222k", fs=2.2222e6, foff=250.0, kind="noise", seed=1100, n=20000))
    # ragged / edge lengths at 2.4 MS/s (decimate needs n>27, filtfilt needs n_dec>15, >=2 symbols)
    for n in [0, 1, 2, 3, 15, 16, 17, 27, 28, 29, 100, 159, 160, 161, 170, 171, 300, 1000, 2047, 2048,
              2049, 4095, 4097, 20481, 65537]:
        cases.append(dict(name=f"edge_2400_n{n}", fs=2.4e6, foff=0.0 if n % 2 else 1171.875,
                          kind="noise", seed=2000 + n, n=n))
    # edge lengths without decimation (fs <= 480 kHz)
    for n in [1, 12, 15, 16, 17, 24, 25, 100, 5000]:
        cases.append(dict(name=f"edge_225_n{n}", fs=0.225e6, foff=500.0, kind="noise", seed=3000 + n, n=n))
    # q=41 stress (memory 8246 samples) with a longer input
    cases.append(dict(name="q41_10M_1M", fs=10e6, foff=1171.875, kind="noise", seed=41, n=1048576))
    # synthetic pi/4-DQPSK (stored as bytes)
    cases.append(dict(name="dqpsk_2400_128k", fs=2.4e6, foff=0.0, kind="dqpsk", seed=1, n=131072, coff=0.0))
    cases.append(dict(name="dqpsk_2400_128k_off", fs=2.4e6, foff=1171.875, kind="dqpsk", seed=2, n=131072,
                      coff=1171.875))
    cases.append(dict(name="dqpsk_1800_64k", fs=1.8e6, foff=0.0, kind="dqpsk", seed=3, n=65536, coff=0.0))
    # constant / zero inputs (max|s| == 0 branch of the slicer, DC steady state)
    cases.append(dict(name="zeros_2400", fs=2.4e6, foff=0.0, kind="const", value=[127, 128], n=4096))
    cases.append(dict(name="dc_2400", fs=2.4e6, foff=1171.875, kind="const", value=[200, 90], n=8192))

    for c in cases:
        if c["kind"] == "noise":
            u8 = synth.noise_cu8(c["n"], c["seed"])
        elif c["kind"] == "dqpsk":
            u8, _ = synth.dqpsk_cu8(c["n"], c["fs"], c["seed"], carrier_offset=c["coff"])
            inputs[c["name"]] = u8
        else:
            u8 = np.tile(np.array(c["value"], dtype=np.uint8), c["n"])
        x = synth.cu8_to_c128(u8)
        p = SignalProcessor(c["fs"])
        hard = p.process(x, c["foff"])
        soft = np.asarray(p.symbols, dtype=np.complex128)
        assert hard.dtype == np.uint8
        proc[c["name"] + "__hard"] = hard
        proc[c["name"] + "__soft"] = soft
        c2 = dict(c)
        c2.update(n_hard=int(len(hard)), n_soft=int(len(soft)), sha256_hard=sha(hard),
                  hist=[int(v) for v in np.bincount(hard, minlength=4)])
        manifest["cases"].append(c2)
        print(f"{c['name']:24s} n={c['n']:8d} -> soft {len(soft):6d} hard {len(hard):6d} {sha(hard)[:16]}")

    # exact (float-typed, not cu8-derived) inputs: complex128 gaussian stored as is, small
    rng = np.random.default_rng(555)
    xg = (rng.standard_normal(6000) + 1j * rng.standard_normal(6000)) * 0.3
    inputs["gauss_c128"] = xg
    p = SignalProcessor(2.4e6)
    proc["gauss_c128__hard"] = p.process(xg, 777.0)
    proc["gauss_c128__soft"] = np.asarray(p.symbols)
    manifest["cases"].append(dict(name="gauss_c128", fs=2.4e6, foff=777.0, kind="c128", n=6000,
                                  n_hard=int(len(proc["gauss_c128__hard"])),
                                  n_soft=int(len(proc["gauss_c128__soft"])),
                                  sha256_hard=sha(proc["gauss_c128__hard"])))

    # C3-style channelised parity: oracle per carrier = p.process(p.frequency_shift(x, f_k))
    offs = [(k - 3.5) * 25000.0 for k in range(8)]
    u8, _ = synth.multicarrier_cu8(65536, 2.4e6, offs, seed0=100)
    inputs["mc8_2400_64k"] = u8
    x = synth.cu8_to_c128(u8)
    for k, f in enumerate(offs):
        p = SignalProcessor(2.4e6)
        hard = p.process(p.frequency_shift(x, f))
        proc[f"mc8_k{k}__hard"] = hard
        proc[f"mc8_k{k}__soft"] = np.asarray(p.symbols)
        manifest["cases"].append(dict(name=f"mc8_k{k}", fs=2.4e6, foff=0.0, pre_shift=f, kind="mc8",
                                      input="mc8_2400_64k", n=65536, n_hard=int(len(hard)),
                                      n_soft=int(len(p.symbols)), sha256_hard=sha(hard)))

    # ------------------------------------------------------------------ per-method goldens
    x = synth.cu8_to_c128(synth.noise_cu8(4000, 77))
    p = SignalProcessor(2.4e6)
    stages["x4000_seed"] = np.array([77])
    stages["filter_default"] = p.filter_signal(x)                      # bandwidth=25000, fs=2.4e6
    stages["filter_bw50k"] = p.filter_signal(x, bandwidth=50000)
    stages["filter_240k"] = p.filter_signal(x, 25000, 240000.0)
    stages["filter_clamp_hi"] = p.filter_signal(x, 25000, 20000.0)     # cutoff clamped to 0.99
    stages["filter_clamp_lo"] = p.filter_signal(x, 100.0, 2.4e6)       # cutoff clamped to 0.01
    stages["filter_short15"] = p.filter_signal(x[:15])                 # filtfilt raises -> input returned
    stages["filter_short16"] = p.filter_signal(x[:16], 25000, 240000.0)
    stages["shift_1000"] = p.frequency_shift(x, 1000)
    stages["shift_m3515_240k"] = p.frequency_shift(x, -3515.625, 240000.0)
    stages["shift_0"] = p.frequency_shift(x, 0)
    stages["extract_default"] = p.extract_symbols(x)                   # sps=133, step=16
    stages["extract_240k"] = p.extract_symbols(x, 240000.0)            # sps=13
    stages["extract_300k"] = p.extract_symbols(x, 300000.0)            # sps=16, step=2
    stages["extract_18k"] = p.extract_symbols(x[:50], 18000.0)         # sps=1 passthrough
    stages["extract_short"] = p.extract_symbols(x[:5], 240000.0)       # n < sps -> empty
    stages["demod_x"] = p.demodulate_dqpsk(x)
    stages["demod_2"] = p.demodulate_dqpsk(x[:2])
    stages["demod_zeros"] = p.demodulate_dqpsk(np.zeros(10, dtype=complex))
    stages["resample_1200k"] = p.resample(x[:1000], 1.2e6)
    stages["resample_up"] = p.resample(x[:301], 3.0e6)
    for q in (7, 10, 41):
        stages[f"decimate_q{q}"] = signal.decimate(x, q)
    # slicer boundary probes: phases exactly at / next to the thresholds
    th = np.array([-5 * np.pi / 8, -3 * np.pi / 8, 3 * np.pi / 8, 5 * np.pi / 8, np.pi, -np.pi, 0.0])
    probe = np.concatenate([th, np.nextafter(th, 10), np.nextafter(th, -10)])
    seq = np.empty(2 * len(probe), dtype=complex)
    seq[0::2] = 1.0
    seq[1::2] = np.exp(1j * probe)
    stages["demod_probe_in"] = seq
    stages["demod_probe"] = p.demodulate_dqpsk(seq)

    # ------------------------------------------------------------------ design tables
    for fs in RATES + [2.2222e6]:
        key = f"{int(fs)}"
        q = int(fs / 240000) if fs > 480000 else 1
        cur = fs / q if q > 1 else fs
        if q > 1:
            sos = signal.cheby1(8, 0.05, 0.8 / q, output="sos")
            design[f"sos_{key}"] = sos
            design[f"soszi_{key}"] = signal.sosfilt_zi(sos)
        cutoff = min(0.99, max(0.01, (25000 / 2) / (cur / 2)))
        b, a = signal.butter(4, cutoff, btype="low")
        design[f"b_{key}"] = b
        design[f"a_{key}"] = a
        design[f"zi_{key}"] = signal.lfilter_zi(b, a)
        design[f"meta_{key}"] = np.array([fs, q, cur, cutoff, int(cur / 18000)])
    for cutoff in (0.01, 0.99, 12500 / 1.2e6):
        b, a = signal.butter(4, cutoff, btype="low")
        design[f"b_w{cutoff:.6f}"] = b
        design[f"a_w{cutoff:.6f}"] = a
        design[f"zi_w{cutoff:.6f}"] = signal.lfilter_zi(b, a)

    np.savez_compressed(os.path.join(HERE, "inputs.npz"), **inputs)
    np.savez_compressed(os.path.join(HERE, "process.npz"), **proc)
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **stages)
    np.savez_compressed(os.path.join(HERE, "design.npz"), **design)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    for fn in ("inputs.npz", "process.npz", "stages.npz", "design.npz", "manifest.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
