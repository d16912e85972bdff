"""tetraear_amd -- MI355X-native IQ -> symbol front end with TetraEar's `tetraear.signal` API.

    from tetraear_amd.signal import SignalProcessor      # drop-in for tetraear.signal.SignalProcessor
    from tetraear_amd.batch import BatchDemodulator      # many carriers per call, device-resident
    from tetraear_amd.batch import batch_demodulator     # ... with three steps in flight (PipelinedBatchDemodulator)

All arithmetic runs in hand-written HIP kernels behind the C-ABI of include/tetrahip.h
(libtetrahip.so, loaded with ctypes).  There is no CPU compute path.
"""


def __getattr__(name):
    if name == "SignalProcessor":
        from tetraear_amd.signal.processor import SignalProcessor
        return SignalProcessor
    if name in ("BatchDemodulator", "PipelinedBatchDemodulator", "batch_demodulator"):
        from tetraear_amd import batch
        return getattr(batch, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["SignalProcessor", "BatchDemodulator", "PipelinedBatchDemodulator", "batch_demodulator"]
__version__ = "0.1.2"
