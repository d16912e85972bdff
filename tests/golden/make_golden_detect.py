#!/usr/bin/env python3
"""Golden vectors for the scanner heuristics by importing the reference's
tetraear.signal.scanner.TetraSignalDetector (importable here: its decoder import is optional).
    python tests/golden/make_golden_detect.py  ->  tests/golden/detect.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tetraear_amd import synth  # noqa: E402   (the reference itself is imported inside main() only)

CASES = [("noise", 7, 30000), ("noise", 8, 999), ("noise", 9, 1300), ("dqpsk", 1, 40000), ("dqpsk", 2, 131072),
         ("dc", 0, 5000), ("noise", 10, 90), ("dqpsk", 3, 2600)]


def make(kind, seed, n):
    if kind == "noise":
        return synth.cu8_to_c128(synth.noise_cu8(n, seed))
    if kind == "dqpsk":
        x, _ = synth.dqpsk_baseband(n, 2.4e6, seed)
        rng = np.random.default_rng(seed + 7)
        return x + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return np.full(n, 0.3 - 0.2j)


def main():
    sys.path.insert(0, "/root/reference")
    from tetraear.signal.scanner import TetraSignalDetector
    det = TetraSignalDetector(2.4e6)
    out = {}
    for i, (kind, seed, n) in enumerate(CASES):
        x = make(kind, seed, n)
        if kind == "dqpsk":
            out[f"x_{i}"] = x                      # libm-dependent input: stored
        p = det.calculate_power(x)
        t, c = det.detect_tetra_modulation(x)
        s, m = det.detect_sync_pattern(x)
        out[f"res_{i}"] = np.array([p, float(t), c, float(s), m])
        print(kind, n, out[f"res_{i}"])
    np.savez_compressed(os.path.join(HERE, "detect.npz"), **out)


if __name__ == "__main__":
    main()
