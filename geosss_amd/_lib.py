"""ctypes binding of libgsss_hip.so (include/gsss.h).  There is no CPU fallback: if the
library is missing or no gfx950 device is visible, every compute entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSSS_HIP_LIB") or os.path.join(_HERE, "libgsss_hip.so")  # env: side-by-side A/B builds

VMF_MIXTURE, BINGHAM, CURVE_VMF, CPD = 1, 2, 3, 4
SHRINK, REJECT, RWMH, HMC, INDEP, MIX = 0, 1, 2, 3, 4, 5
MODE_EXACT, MODE_FAST = 0, 1
VARIANT_FAST_DOUBLE = 100
VARIANT_FAST_VERIFY = 101
CHAIN_MAX_TRIES, CHAIN_NONFINITE, CHAIN_REPLAY_EXHAUSTED, CHAIN_COUNTER_SATURATED = 1, 2, 4, 8
ABI_VERSION = 10
STATS_NO_SECOND_MOMENT = 1


class GsssError(RuntimeError):
    pass


class TargetDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("k", C.c_int32), ("reserved", C.c_int32),
                ("mu", C.c_void_p), ("logc", C.c_void_p), ("A", C.c_void_p), ("knots", C.c_void_p),
                ("kappa", C.c_double),
                ("source", C.c_void_p), ("source_w", C.c_void_p), ("target", C.c_void_p), ("target_w", C.c_void_p),
                ("n_target", C.c_int32), ("target_dim", C.c_int32), ("k_nn", C.c_int32), ("outlier", C.c_int32),
                ("sigma", C.c_double), ("beta", C.c_double), ("omega", C.c_double), ("log_volume", C.c_double)]


class RunArgs(C.Structure):
    _fields_ = [("state_dev", C.c_void_p), ("samples_dev", C.c_void_p), ("n_reject_dev", C.c_void_p),
                ("n_tries_dev", C.c_void_p), ("err_dev", C.c_void_p), ("replay_dev", C.c_void_p),
                ("replay_stride", C.c_int64), ("n_chains", C.c_int64), ("n_steps", C.c_int64), ("thin", C.c_int64),
                ("seed", C.c_uint64), ("chain_offset", C.c_uint64), ("step_offset", C.c_uint64),
                ("sampler", C.c_int32), ("mode", C.c_int32), ("max_tries", C.c_int32), ("variant", C.c_int32),
                ("rng_state_dev", C.c_void_p), ("samples_chain_rows", C.c_int64), ("placement", C.c_int32),
                ("stats_lags", C.c_int32), ("stats_dev", C.c_void_p), ("stats_dirs_dev", C.c_void_p),
                ("stats_modes", C.c_int32), ("n_leapfrog", C.c_int32), ("stepsize_dev", C.c_void_p),
                ("n_accept_dev", C.c_void_p), ("momenta_dev", C.c_void_p), ("adapt_steps", C.c_int64),
                ("mixing_probability", C.c_double), ("adapt_left_dev", C.c_void_p), ("n_rwmh_dev", C.c_void_p),
                ("momenta_samples_dev", C.c_void_p), ("stepsize_trace_dev", C.c_void_p), ("stats_flags", C.c_int32),
                ("reserved0", C.c_int32)]


# symbol -> (restype, argtypes); must list every function include/gsss.h declares
SIGNATURES = {
    "gsss_abi_version": (C.c_int, []),
    "gsss_last_error": (C.c_char_p, []),
    "gsss_source_digest": (C.c_char_p, []),
    "gsss_device_count": (C.c_int, []),
    "gsss_target_create": (C.c_int, [C.POINTER(TargetDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "gsss_target_destroy": (C.c_int, [C.c_void_p]),
    "gsss_target_dim": (C.c_int, [C.c_void_p]),
    "gsss_logprob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gsss_gradient": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gsss_run": (C.c_int, [C.c_void_p, C.POINTER(RunArgs), C.c_void_p]),
    "gsss_stats_rows": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gsss_last_launch": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "gsss_mode_supported": (C.c_int, [C.c_void_p, C.c_int32]),
    "gsss_variant_name": (C.c_char_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "gsss_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "gsss_sample_sphere": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]),
    "gsss_tangent_s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]),
    "gsss_screen_constants": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "gsss_f32_error_sweep": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int, C.c_void_p]),
    "gsss_rows_to_components": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p]),
    "gsss_components_to_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_void_p]),
    "gsss_samples_to_chains": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int, C.c_void_p]),
    "gsss_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "gsss_free": (C.c_int, [C.c_void_p, C.c_int]),
    "gsss_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gsss_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gsss_malloc_host": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "gsss_free_host": (C.c_int, [C.c_void_p]),
    "gsss_host_register": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    "gsss_host_unregister": (C.c_int, [C.c_void_p]),
    "gsss_memcpy_d2h_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gsss_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p]),
    "gsss_stream_synchronize": (C.c_int, [C.c_int, C.c_void_p]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for this) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsssError(f"{LIB_PATH} is missing: build it with `python -m geosss_amd.build` "
                        "(there is no CPU fallback for the sampler)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gsss_abi_version() != ABI_VERSION:
        raise GsssError(f"libgsss_hip.so has ABI {lib.gsss_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().gsss_last_error().decode("utf-8", "replace")
        if code in (-1, -2):
            raise ValueError(f"gsss: {msg} (code {code})")
        raise GsssError(f"gsss: {msg} (code {code})")


def require_device():
    lib = load()
    n = lib.gsss_device_count()
    if n <= 0:
        raise GsssError("no HIP device visible: the geodesic slice sampler runs on MI355X only "
                        "(no CPU fallback is provided)")
    return n
