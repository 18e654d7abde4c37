"""ctypes binding of libhadi.so (the C ABI declared in include/hadi.h).

The library is the product: if it is missing or no gfx950 GPU is usable, every compute entry
point raises -- there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhadi.so")

HADI_OK = 0
EU, AM, DIV, AM_DIV = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
SCHEME_DOUGLAS, SCHEME_CRAIG_SNEYD = 0, 1
STATE_FP64, STATE_FP32 = 0, 1
CALL, PUT = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class HadiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libhadi error %d: %s" % (status, message))
        self.status = status


class Problem(C.Structure):
    """struct hadi_problem (include/hadi.h)."""
    _fields_ = [
        ("n_instances", C.c_int), ("m1", C.c_int), ("m2", C.c_int), ("variant", C.c_int), ("memspace", C.c_int),
        ("N", C.c_int), ("delta_t", C.c_double), ("theta", C.c_double),
        ("N_i", _ip), ("delta_t_i", _dp),
        ("r_d", C.c_double), ("r_f", C.c_double),
        ("rho", C.c_double), ("sigma", C.c_double), ("kappa", C.c_double), ("eta", C.c_double),
        ("rho_i", _dp), ("sigma_i", _dp), ("kappa_i", _dp), ("eta_i", _dp),
        ("vec_s", _dp), ("vec_v", _dp), ("delta_s", _dp), ("delta_v", _dp),
        ("num_dividends", C.c_int),
        ("dividend_dates", _dp), ("dividend_amounts", _dp), ("dividend_percentages", _dp),
        ("U", _dp), ("U_0", _dp), ("lambda_bar", _dp),
        ("scheme", C.c_int),
        ("state_precision", C.c_int),
        ("option_type", C.c_int),
        ("strike_i", _dp),
        ("V_0_i", _dp),
    ]


class Timing(C.Structure):
    """struct hadi_timing (include/hadi.h)."""
    _fields_ = [
        ("setup_ms", C.c_double), ("sweep_ms", C.c_double), ("finish_ms", C.c_double),
        ("pass_a_ms", C.c_double), ("pass_b_ms", C.c_double),
        ("pass_a_launches", C.c_longlong), ("pass_b_launches", C.c_longlong),
        ("point_steps", C.c_longlong),
    ]


# every symbol include/hadi.h declares (tests check the library exports all of them)
EXPORTS = [
    "hadi_create", "hadi_destroy", "hadi_last_error", "hadi_status_string", "hadi_version",
    "hadi_set_profiling", "hadi_get_timing", "hadi_set_tuning", "hadi_get_tuning", "hadi_device_info", "hadi_describe_last_sweep",
    "hadi_stream", "hadi_wait_stream",
    "hadi_make_grid", "hadi_rebuild_variance", "hadi_find_s_index", "hadi_find_v_index",
    "hadi_DO_timestepping", "hadi_parallel_DO_solve",
    "hadi_compute_base_prices", "hadi_compute_base_prices_american",
    "hadi_compute_base_prices_dividends", "hadi_compute_base_prices_american_dividends",
    "hadi_compute_jacobian", "hadi_compute_jacobian_american",
    "hadi_compute_jacobian_dividends", "hadi_compute_jacobian_american_dividends",
    "hadi_lm_partials", "hadi_lm_partials_device", "hadi_lm_solve", "hadi_compute_parameter_update",
    "hadi_debug_row_pass", "hadi_debug_col_solve", "hadi_debug_rcp",
]

_lib = None
_others = {}


def lib(path=None):
    """Loads libhadi.so; raises if it has not been built (python __graft_entry__.py build).  `path`: another build of
    the same ABI (tests load the checking build libhadi_strict.so next to the product)."""
    global _lib
    if path is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        key = os.path.abspath(path)
        if key not in _others:
            _others[key] = _load(key)
        return _others[key]
    if _lib is not None:
        return _lib
    _lib = _load(LIB_PATH)
    return _lib


def _load(LIB_PATH):
    if not os.path.exists(LIB_PATH):
        raise HadiError(-1, "libhadi.so not built at %s -- run `python __graft_entry__.py` (hipcc, gfx950); "
                            "this package has no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 under the same soname as
    # /opt/rocm's.  Whichever is loaded first serves both; torch cannot see the GPU through the system
    # copy, so when torch is installed let it load first (libhadi runs fine on torch's copy).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.hadi_last_error.restype = C.c_char_p
    L.hadi_status_string.restype = C.c_char_p
    L.hadi_stream.restype = C.c_void_p
    L.hadi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.hadi_destroy.argtypes = [C.c_void_p]
    L.hadi_last_error.argtypes = [C.c_void_p]
    L.hadi_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.hadi_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.hadi_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.hadi_get_tuning.argtypes = [C.c_void_p, C.c_char_p, _ip]
    L.hadi_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.hadi_lm_partials_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp]
    L.hadi_debug_row_pass.argtypes = [C.c_void_p, C.POINTER(Problem), C.c_int, C.c_void_p]
    L.hadi_debug_col_solve.argtypes = [C.c_void_p, C.POINTER(Problem), C.c_void_p]
    L.hadi_debug_rcp.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.hadi_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _ip, C.c_char_p, C.c_int]
    L.hadi_describe_last_sweep.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.hadi_stream.argtypes = [C.c_void_p]
    L.hadi_make_grid.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
    L.hadi_rebuild_variance.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
    L.hadi_find_s_index.argtypes = [C.c_int, _dp, C.c_double]
    L.hadi_find_v_index.argtypes = [C.c_int, _dp, C.c_double]
    L.hadi_DO_timestepping.argtypes = [C.c_void_p, C.POINTER(Problem)]
    L.hadi_parallel_DO_solve.argtypes = [C.c_void_p, C.POINTER(Problem), C.c_double, C.c_double, C.c_void_p]
    for sfx in ("", "_american", "_dividends", "_american_dividends"):
        getattr(L, "hadi_compute_base_prices" + sfx).argtypes = [
            C.c_void_p, C.POINTER(Problem), C.c_double, C.c_double, C.c_void_p]
        getattr(L, "hadi_compute_jacobian" + sfx).argtypes = [
            C.c_void_p, C.POINTER(Problem), C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.hadi_lm_partials.argtypes = [C.c_int, _dp, _dp, _dp]
    L.hadi_lm_solve.argtypes = [_dp, C.c_double, _dp]
    L.hadi_compute_parameter_update.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp]
    return L
