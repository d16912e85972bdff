"""Drop-in for `tetraear.signal` restricted to the demodulation path.

Only `SignalProcessor` is provided; capture and scanner classes stay with the reference
(tetraear/signal/__init__.py:11-32 exports them lazily in the same way).
"""


def __getattr__(name):
    if name == "SignalProcessor":
        from tetraear_amd.signal.processor import SignalProcessor
        return SignalProcessor
    raise AttributeError(f"module {__name__!r} has no attribute {name!r} "
                         "(only the SignalProcessor path is replaced; see INTEGRATION.md)")


__all__ = ["SignalProcessor"]
