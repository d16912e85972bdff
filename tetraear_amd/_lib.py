"""ctypes binding of libtetrahip.so (include/tetrahip.h).

The library is built in-tree (tetraear_amd/csrc/Makefile, hipcc --offload-arch=gfx950) and there is
deliberately no fallback: if it is missing, or no MI355X is visible, the error is raised to the
caller.  Nothing in this package computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TETRAHIP_LIB", os.path.join(_HERE, "libtetrahip.so"))  # override: experiments only

FMT_CU8, FMT_CS8, FMT_CF32, FMT_CF64 = 0, 1, 2, 3
MODE_REFERENCE, MODE_TETRA, MODE_TETRA_GARDNER = 0, 1, 2
FMT_BYTES = {FMT_CU8: 2, FMT_CS8: 2, FMT_CF32: 8, FMT_CF64: 16}


class TetraHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libtetrahip error {code}: {text}")
        self.code = code


class PlanInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("rate_dec", C.c_double), ("n_samples", C.c_int64),
                ("n_dec", C.c_int64), ("n_carriers", C.c_int32), ("q", C.c_int32), ("sps", C.c_int32),
                ("phase_step", C.c_int32), ("max_soft", C.c_int32), ("lpf_applied", C.c_int32),
                ("in_fmt", C.c_int32), ("mode", C.c_int32), ("device", C.c_int32), ("dec_engine", C.c_int32),
                ("gardner_segments", C.c_int32)]


_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/tetrahip.h declares
SIGNATURES = {
    "tdm_version": (C.c_int, []),
    "tdm_device_count": (C.c_int, []),
    "tdm_last_error": (C.c_int, [C.c_char_p, _sz]),
    "tdm_debug_set": (C.c_int, [C.c_char_p, _i64]),
    "tdm_debug_get": (C.c_int, [C.c_char_p, _P(_i64)]),
    "tdm_plan_create": (C.c_int, [_f64, _i64, _i32, _i32, _i32, _i32, _P(_vp)]),
    "tdm_plan_destroy": (C.c_int, [_vp]),
    "tdm_plan_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "tdm_gardner_geometry": (C.c_int, [C.c_double, _i64, C.c_int32, _vp]),
    "tdm_plan_get_info": (C.c_int, [_vp, _P(PlanInfo)]),
    "tdm_plan_resize": (C.c_int, [_vp, _i64]),
    "tdm_process": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tdm_process_device": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tdm_plan_sync": (C.c_int, [_vp]),
    "tdm_plan_wait_for": (C.c_int, [_vp, _vp]),
    "tdm_process_pipelined": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tdm_filter_signal": (C.c_int, [_vp, _i64, _f64, _f64, _vp, _P(_i32), _i32]),
    "tdm_frequency_shift": (C.c_int, [_vp, _i64, _f64, _f64, _vp, _i32]),
    "tdm_extract_symbols": (C.c_int, [_vp, _i64, _f64, _f64, _vp, _P(_i64), _P(_i32), _i32]),
    "tdm_demodulate_dqpsk": (C.c_int, [_vp, _i64, _vp, _P(_i64), _P(_f64), _i32]),
    "tdm_set_stream": (C.c_int, [_vp]),
    "tdm_plan_stream": (C.c_int, [_vp, _P(_vp)]),
    "tdm_plan_rrc_filter": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "tdm_decimate": (C.c_int, [_vp, _i64, _i32, _vp, _P(_i64), _i32]),
    "tdm_resample": (C.c_int, [_vp, _i64, _i64, _vp, _i32]),
    "tdm_spectrum_gate": (C.c_int, [_vp, _i32, _i64, _i64, _i32, _f64, _vp, _vp, _i32, _i32]),
    "tdm_detect": (C.c_int, [_vp, _i64, _i32, _f64, _vp, _i32]),
    "tdm_find_sync": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _i32, _i32]),
    "tdm_channelise": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _P(_i64), _i32, _i32]),
    "tdm_channelise_batch": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _i64, _P(_i64), _i32, _i32]),
    "tdm_occupancy_gate": (C.c_int, [_vp, _i64, _i32, _i32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "tdm_process_device_rows": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tdm_hbm_ceiling": (C.c_int, [_i32, _sz, _i32, _P(_f64)]),
    "tdm_host_register": (C.c_int, [_i32, _vp, _sz]),
    "tdm_host_unregister": (C.c_int, [_i32, _vp]),
    "tdm_dev_alloc": (C.c_int, [_i32, _sz, _P(_vp)]),
    "tdm_dev_free": (C.c_int, [_i32, _vp]),
    "tdm_dev_upload": (C.c_int, [_i32, _vp, _vp, _sz]),
    "tdm_dev_download": (C.c_int, [_i32, _vp, _vp, _sz]),
    "tdm_dev_sync": (C.c_int, [_i32]),
    "tdm_plan_time_begin": (C.c_int, [_vp]),
    "tdm_plan_time_begin_total": (C.c_int, [_vp]),
    "tdm_plan_time_end": (C.c_int, [_vp, _P(C.c_float)]),
    "tdm_plan_stage_times": (C.c_int, [_vp, _i32, _P(C.c_char_p), _P(C.c_float), _P(_i32)]),
    "tdm_design_dump": (C.c_int, [_f64, _i64, _vp, _vp, _vp, _vp, _vp, _P(_i32), _P(_f64)]),
    "tdm_design_butter": (C.c_int, [_f64, _f64, _vp, _vp, _vp]),
}

_lib = None

# The header version these bindings (PlanInfo's layout, SIGNATURES) were written for: include/tetrahip.h TDM_VERSION.
# tests/test_abi_cpu.py holds it to the header; load() holds the library to it.
ABI_VERSION = 102
ALLOW_EXPERIMENT_ENV = "TETRAHIP_ALLOW_EXPERIMENT"   # timing-only builds (negative version): tools/ab_*.py only


def header_version(path=None):
    """TDM_VERSION as include/tetrahip.h states it (the header travels with the repo; a deployment without it has
    ABI_VERSION)."""
    path = path or os.path.join(os.path.dirname(_HERE), "include", "tetrahip.h")
    with open(path) as f:
        for line in f:
            if line.startswith("#define TDM_VERSION"):
                return int(line.split()[2])
    raise TetraHipError(-2, f"no TDM_VERSION in {path}")


def load():
    """Load libtetrahip.so; raises (never falls back) when it is absent, when it is not the version these bindings were
    written for (a stale .so fills a shorter tdm_plan_info), or when it is a timing-only experiment build (negative
    version: such a library knowingly returns wrong results)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TetraHipError(-2, f"{LIB_PATH} not built (run `make -C tetraear_amd/csrc`); "
                                    "this package has no CPU path")
        lib = C.CDLL(LIB_PATH)
        lib.tdm_version.restype = C.c_int
        lib.tdm_version.argtypes = []
        v = lib.tdm_version()
        if v < 0 and not os.environ.get(ALLOW_EXPERIMENT_ENV):
            raise TetraHipError(-2, f"{LIB_PATH} is a timing-only experiment build (tdm_version() = {v}): its results "
                                    f"are knowingly wrong; set {ALLOW_EXPERIMENT_ENV}=1 only to time it")
        if abs(v) != ABI_VERSION:
            raise TetraHipError(-2, f"{LIB_PATH} reports tdm_version() = {v}, these bindings are for {ABI_VERSION} "
                                    "(rebuild: `make -C tetraear_amd/csrc`)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    load().tdm_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        raise TetraHipError(rc, last_error())
    return rc


class debug_option:
    """`with debug_option("row_walk", 1): ...` -- set a tdm_debug_set switch for the duration of a block (tests and
    experiments; include/tetrahip.h lists the switches)."""

    def __init__(self, key, value):
        self.key, self.value = key.encode(), int(value)

    def __enter__(self):
        lib = load()
        old = C.c_int64(0)
        check(lib.tdm_debug_get(self.key, C.byref(old)))
        self.old = old.value
        check(lib.tdm_debug_set(self.key, self.value))
        return self

    def __exit__(self, *exc):
        check(load().tdm_debug_set(self.key, self.old))
        return False


def ptr(a):
    """numpy array (or None) -> void*"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
