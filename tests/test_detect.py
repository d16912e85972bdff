"""Scanner heuristics: kernel body (CPU emulation) and GPU path against golden values
produced by importing the reference's TetraSignalDetector (tests/golden/make_golden_detect.py)."""
import os

import numpy as np
import pytest

from tests.golden.make_golden_detect import CASES, make

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "detect.npz"))


def _input(i):
    kind, seed, n = CASES[i]
    return G[f"x_{i}"] if kind == "dqpsk" else make(kind, seed, n)


def _check(o, ref):
    assert abs(o[0] - ref[0]) < 1e-9                     # power (dB)
    assert o[1] == ref[1] and abs(o[2] - ref[2]) < 2e-4   # is_tetra, confidence (atan2 last-bit effects)
    assert o[3] == ref[3] and abs(o[4] - ref[4]) < 1e-12  # found_sync, max correlation (k/31)


def test_emul_detect_matches_reference():
    from tests.emul import emul
    for i in range(len(CASES)):
        _check(emul.detect(_input(i), 2.4e6), G[f"res_{i}"])


@pytest.mark.gpu
def test_gpu_detector_matches_reference():
    from tetraear_amd.detector import TetraSignalDetector
    det = TetraSignalDetector(2.4e6)
    for i in range(len(CASES)):
        x, ref = _input(i), G[f"res_{i}"]
        p = det.calculate_power(x)
        t, c = det.detect_tetra_modulation(x)
        s, m = det.detect_sync_pattern(x)
        _check(np.array([p, float(t), c, float(s), m]), ref)
    assert det.calculate_power(np.array([])) == -85.0     # scanner.py:51-52
