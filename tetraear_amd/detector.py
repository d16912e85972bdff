"""GPU version of the scanner's signal heuristics, same method names and
return conventions as `tetraear.signal.scanner.TetraSignalDetector` (scanner.py:24-147)."""
import numpy as np

from tetraear_amd import _lib
from tetraear_amd._lib import check, ptr


class TetraSignalDetector:
    def __init__(self, sample_rate=2.4e6, noise_floor=-45, bottom_threshold=-85, device=0):
        self.sample_rate = sample_rate
        self.symbol_rate = 18000
        self.channel_bandwidth = 25000
        self.noise_floor = noise_floor
        self.bottom_threshold = bottom_threshold
        self.device = device

    def _run(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.complex128)
        out = np.zeros(8)
        check(_lib.load().tdm_detect(ptr(x), len(x), 1, float(self.sample_rate), ptr(out), self.device))
        return out

    def calculate_power(self, samples):
        if np.asarray(samples).size == 0:
            return float(self.bottom_threshold)
        return float(self._run(samples)[0])

    def detect_tetra_modulation(self, samples):
        if len(samples) < 1000:
            return False, 0.0
        o = self._run(samples)
        return bool(o[1] != 0.0), float(o[2])

    def detect_sync_pattern(self, samples):
        o = self._run(samples)
        return bool(o[3] != 0.0), float(o[4])
