"""Host-call time of every public method of tetraear_amd.signal.SignalProcessor on a capture-sized array (262 144 samples; host
arrays in and out, PCIe and the ctypes shim included).  usage (GPU box): python tools/methods_time.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from tetraear_amd import synth
from tetraear_amd.signal import SignalProcessor
p = SignalProcessor(2.4e6)
x = synth.cu8_to_c128(synth.noise_cu8(262144, 3))
def t(name, f, reps=5):
    f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    print(f"{name:28s} {(time.perf_counter()-t0)/reps*1e3:8.2f} ms  -> {len(r)}")
t("filter_signal", lambda: p.filter_signal(x))
t("filter_signal 240k", lambda: p.filter_signal(x[:26215], 25000, 240000.0))
t("frequency_shift", lambda: p.frequency_shift(x, 1000.0))
t("extract_symbols 240k", lambda: p.extract_symbols(x[:26215], 240000.0))
t("extract_symbols 2.4M", lambda: p.extract_symbols(x))
t("demodulate_dqpsk 2016", lambda: p.demodulate_dqpsk(x[:2016]))
t("demodulate_dqpsk 262144", lambda: p.demodulate_dqpsk(x))
t("resample 262144->131072", lambda: p.resample(x, 1.2e6))
t("process", lambda: p.process(x, 1171.875))
