"""tetraear_amd -- MI355X-native IQ -> symbol front end with TetraEar's `tetraear.signal` API.

    from tetraear_amd.signal import SignalProcessor      # drop-in for tetraear.signal.SignalProcessor
    from tetraear_amd.batch import BatchDemodulator      # many carriers per call, device-resident

All arithmetic runs in hand-written HIP kernels behind the C-ABI of include/tetrahip.h
(libtetrahip.so, loaded with ctypes).  There is no CPU compute path.
"""


def __getattr__(name):
    if name == "SignalProcessor":
        from tetraear_amd.signal.processor import SignalProcessor
        return SignalProcessor
    if name == "BatchDemodulator":
        from tetraear_amd.batch import BatchDemodulator
        return BatchDemodulator
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["SignalProcessor", "BatchDemodulator"]
__version__ = "0.1.0"
