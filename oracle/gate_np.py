"""ORACLE (test infrastructure): numpy restatement of the spectrum / AFC / signal gate that sits in
front of process() in the reference's capture loop, tetraear/ui/modern.py:1921-2021 (inside
CaptureThread.run; the module needs PyQt6 and cannot be imported here, and the block is inline code,
not a function).  Pinned by tests/golden/gate.npz: tests/golden/make_golden_gate.py takes the block's
statements out of the reference's AST and executes them unchanged; this restatement reproduces those
outputs bit for bit (tests/test_gate.py::test_oracle_matches_reference_block)."""
import numpy as np


def gate(samples, sample_rate):
    """Returns dict(peak_freq_offset, signal_power, peak_power, noise_floor, snr, strong, afc)."""
    n_fft = 2048
    if len(samples) < n_fft:
        return dict(peak_freq_offset=0.0, signal_power=0.0, peak_power=0.0, noise_floor=0.0, snr=0.0, strong=False, afc=0.0)
    fft_samples = samples[:n_fft]
    window = np.hanning(n_fft)
    fft = np.fft.fftshift(np.fft.fft(fft_samples * window))
    freqs = np.fft.fftshift(np.fft.fftfreq(n_fft, 1 / sample_rate))
    power = 20 * np.log10(np.abs(fft) / n_fft + 1e-20)
    center_idx = len(power) // 2
    freq_resolution = sample_rate / n_fft
    bandwidth_bins = int(25000 / freq_resolution)
    start_idx = max(0, center_idx - bandwidth_bins // 2)
    end_idx = min(len(power), center_idx + bandwidth_bins // 2)
    signal_power = np.mean(power[start_idx:end_idx])
    peak_power = np.max(power[start_idx:end_idx])
    peak_idx = start_idx + np.argmax(power[start_idx:end_idx])
    peak_freq_offset = freqs[peak_idx]
    noise_bins_end = max(0, start_idx - 10)
    noise_bins_start2 = min(len(power), end_idx + 10)
    noise_list = []
    if noise_bins_end > 0:
        noise_list.extend(power[0:noise_bins_end])
    if len(power) > noise_bins_start2:
        noise_list.extend(power[noise_bins_start2:len(power)])
    noise_floor = np.mean(noise_list) if noise_list else -100
    snr = signal_power - noise_floor
    strong = bool(snr > 15 and peak_power > -70 and (peak_power - signal_power) > 3)
    afc = peak_freq_offset if strong and peak_power > -70 else 0
    return dict(peak_freq_offset=float(peak_freq_offset), signal_power=float(signal_power), peak_power=float(peak_power),
                noise_floor=float(noise_floor), snr=float(snr), strong=strong, afc=float(afc))
