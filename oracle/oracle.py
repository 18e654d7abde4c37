"""ctypes binding of the CPU oracle (oracle/heston_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libheston_oracle.so")
_LIB_XP_PATH = os.path.join(_HERE, "libheston_oracle_xp.so")  # the extended-precision adjudicator (heston_oracle_xp.c)

EU, AM, DIV, AM_DIV = 0, 1, 2, 3
_dp = C.POINTER(C.c_double)


class _Params(C.Structure):
    _fields_ = [
        ("m1", C.c_int), ("m2", C.c_int), ("N", C.c_int), ("variant", C.c_int),
        ("delta_t", C.c_double), ("theta", C.c_double),
        ("r_d", C.c_double), ("r_f", C.c_double),
        ("rho", C.c_double), ("sigma", C.c_double), ("kappa", C.c_double), ("eta", C.c_double),
        ("num_dividends", C.c_int),
        ("div_dates", _dp), ("div_amounts", _dp), ("div_percentages", _dp),
        ("scheme", C.c_int),
        ("state_fp32", C.c_int),
        ("option_type", C.c_int),
        ("strike", C.c_double),
        ("strike_i", _dp),
    ]


class _Dump(C.Structure):
    _fields_ = [("step", C.c_int)] + [(k, _dp) for k in (
        "b", "b1", "b2", "A0U", "A1U", "A2U", "Y0rhs", "Y1", "Y1rhs", "Unext")]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    import fcntl
    fd = os.open(os.path.join(_HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:  # several test ranks may get here at once
        fcntl.flock(fd, fcntl.LOCK_EX)
        srcs = [os.path.join(_HERE, f) for f in ("heston_oracle.c", "heston_oracle.h", "heston_oracle_xp.c", "Makefile")]
        newest = max(os.path.getmtime(f) for f in srcs)
        if force or any(not os.path.exists(l) or os.path.getmtime(l) < newest for l in (_LIB_PATH, _LIB_XP_PATH)):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ho_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


CALL, PUT = 0, 1


def make_params(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, variant=EU,
                dividends=None, scheme=0, state_fp32=0, option_type=CALL, strikes=None):
    """strikes: scalar or per-instance array, needed for option_type=PUT only (boundary value K e^{-r_d t})."""
    p = _Params()
    p.m1, p.m2, p.N, p.variant = m1, m2, N, variant
    p.delta_t, p.theta, p.r_d, p.r_f = delta_t, theta, r_d, r_f
    p.rho, p.sigma, p.kappa, p.eta = rho, sigma, kappa, eta
    p.scheme = scheme
    p.state_fp32 = state_fp32
    p.option_type = option_type
    keep = []
    if strikes is not None:
        ks = _f64(np.atleast_1d(strikes))
        keep.append(ks)
        p.strike = float(ks[0])
        p.strike_i = _p(ks)
    if dividends is not None:
        dates, amounts, pcts = (_f64(x) for x in dividends)
        keep += [dates, amounts, pcts]
        p.num_dividends = len(dates)
        p.div_dates, p.div_amounts, p.div_percentages = _p(dates), _p(amounts), _p(pcts)
    else:
        p.num_dividends = 0
    p._keep = keep
    return p


def grid(m1, S, S_0, K, c, m2, V, V_0, d):
    vs, vv = np.empty(m1 + 1), np.empty(m2 + 1)
    ds, dv = np.empty(m1), np.empty(m2)
    lib().ho_grid(C.c_int(m1), C.c_double(S), C.c_double(S_0), C.c_double(K), C.c_double(c),
                  C.c_int(m2), C.c_double(V), C.c_double(V_0), C.c_double(d),
                  _p(vs), _p(vv), _p(ds), _p(dv))
    return vs, vv, ds, dv


def rebuild_variance(m2, V_0_new, V=5.0, d=5.0 / 500):
    vv, dv = np.empty(m2 + 1), np.empty(m2)
    lib().ho_rebuild_variance(C.c_int(m2), C.c_double(V_0_new), C.c_double(V), C.c_double(d), _p(vv), _p(dv))
    return vv, dv


def find_s_index(vec_s, S_0):
    return lib().ho_find_s_index(C.c_int(len(vec_s) - 1), _p(_f64(vec_s)), C.c_double(S_0))


def find_v_index(vec_v, V_0):
    return lib().ho_find_v_index(C.c_int(len(vec_v) - 1), _p(_f64(vec_v)), C.c_double(V_0))


def solve(params, vec_s, vec_v, delta_s, delta_v, U, U_0=None, dump_step=0):
    """One instance.  Returns (U_T, lambda_bar or None, dump dict or None)."""
    m = (params.m1 + 1) * (params.m2 + 1)
    U = _f64(U).copy()
    U_0 = _f64(U_0)
    lam = np.zeros(m) if params.variant in (AM, AM_DIV) else None
    dump, arrs = None, None
    if dump_step:
        dump = _Dump()
        dump.step = dump_step
        arrs = {}
        for k, _ in _Dump._fields_[1:]:
            arrs[k] = np.zeros(m)
            setattr(dump, k, _p(arrs[k]))
    rc = lib().ho_solve(C.byref(params), _p(_f64(vec_s)), _p(_f64(vec_v)), _p(_f64(delta_s)), _p(_f64(delta_v)),
                        _p(U), _p(U_0), _p(lam), C.byref(dump) if dump is not None else None)
    if rc != 0:
        raise RuntimeError("ho_solve failed rc=%d" % rc)
    return U, lam, arrs


def solve_batch(params, vec_s, vec_v, delta_s, delta_v, U, U_0=None, threads=0, want_lambda=False):
    n = vec_s.shape[0]
    U = _f64(U).copy()
    U_0 = _f64(U_0)
    lam = np.zeros_like(U) if (want_lambda and params.variant in (AM, AM_DIV)) else None
    t = lib().ho_solve_batch(C.byref(params), C.c_int(n), _p(_f64(vec_s)), _p(_f64(vec_v)),
                             _p(_f64(delta_s)), _p(_f64(delta_v)), _p(U), _p(U_0), _p(lam), C.c_int(threads))
    return U, lam, t


_lib_xp = None


def solve_xp(params, vec_s, vec_v, delta_s, delta_v, U, U_0=None, strike=None):
    """ONE instance through the adjudicator: the oracle's source compiled with binary128 arithmetic
    (heston_oracle_xp.c); fp64 inputs widened exactly, outputs rounded to fp64 once.  Returns (U_T, lambda_bar or None).
    ~100x slower than solve(): meant for the handful of instances a parity test or the fuzz tool needs adjudicated."""
    global _lib_xp
    if _lib_xp is None:
        build()
        _lib_xp = C.CDLL(_LIB_XP_PATH)
        _lib_xp.hoxp_solve.restype = C.c_int
    p = params
    m = (p.m1 + 1) * (p.m2 + 1)
    U = _f64(U).reshape(-1).copy()
    U_0 = None if U_0 is None else _f64(U_0).reshape(-1)
    assert U.size == m and np.asarray(vec_s).size == p.m1 + 1 and np.asarray(vec_v).size == p.m2 + 1
    lam = np.zeros(m) if p.variant in (AM, AM_DIV) else None
    nd = p.num_dividends
    rc = _lib_xp.hoxp_solve(
        C.c_int(p.m1), C.c_int(p.m2), C.c_int(p.N), C.c_int(p.variant), C.c_double(p.delta_t), C.c_double(p.theta),
        C.c_double(p.r_d), C.c_double(p.r_f), C.c_double(p.rho), C.c_double(p.sigma), C.c_double(p.kappa), C.c_double(p.eta),
        C.c_int(nd), p.div_dates if nd else None, p.div_amounts if nd else None, p.div_percentages if nd else None,
        C.c_int(p.scheme), C.c_int(p.state_fp32), C.c_int(p.option_type), C.c_double(p.strike if strike is None else strike),
        _p(_f64(vec_s)), _p(_f64(vec_v)), _p(_f64(delta_s)), _p(_f64(delta_v)), _p(U), _p(U_0), _p(lam))
    if rc != 0:
        raise RuntimeError("hoxp_solve failed rc=%d" % rc)
    return U, lam


def base_prices(params, S_0, V_0, vec_s, vec_v, delta_s, delta_v, U, U_0=None, V=5.0, d=5.0 / 500, threads=0):
    n = vec_s.shape[0]
    U = _f64(U).copy()
    vec_v, delta_v = _f64(vec_v).copy(), _f64(delta_v).copy()
    out = np.empty(n)
    rc = lib().ho_base_prices(C.byref(params), C.c_int(n), C.c_double(S_0), C.c_double(V_0), C.c_double(V),
                              C.c_double(d), _p(_f64(vec_s)), _p(vec_v), _p(_f64(delta_s)), _p(delta_v),
                              _p(U), _p(_f64(U_0)), _p(out), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("ho_base_prices failed rc=%d" % rc)
    return out, U


def jacobian(params, S_0, V_0, vec_s, vec_v, delta_s, delta_v, U_0, eps=1e-6, V=5.0, d=5.0 / 500, threads=0):
    n = vec_s.shape[0]
    vec_v, delta_v = _f64(vec_v).copy(), _f64(delta_v).copy()
    J, base = np.empty((n, 5)), np.empty(n)
    rc = lib().ho_jacobian(C.byref(params), C.c_int(n), C.c_double(S_0), C.c_double(V_0), C.c_double(V),
                           C.c_double(d), _p(_f64(vec_s)), _p(vec_v), _p(_f64(delta_s)), _p(delta_v),
                           _p(_f64(U_0)), _p(J), _p(base), C.c_double(eps), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("ho_jacobian failed rc=%d" % rc)
    return J, base


def lm_update(J, residuals, lam):
    J, r = _f64(J), _f64(residuals)
    delta = np.empty(5)
    lib().ho_lm_update(C.c_int(J.shape[0]), _p(J), _p(r), C.c_double(lam), _p(delta))
    return delta


def operator(params, which, vec_s, vec_v, delta_s, delta_v, x, b=None):
    """The reference's per-operator drivers: which = 0/1/2 -> (A_which x, (I - theta dt A_which)^{-1} b or None)."""
    m = (params.m1 + 1) * (params.m2 + 1)
    x = _f64(x)
    res = np.zeros(m)
    sol = np.zeros(m) if (b is not None and which > 0) else None
    rc = lib().ho_operator(C.byref(params), C.c_int(which), _p(_f64(vec_s)), _p(_f64(vec_v)), _p(_f64(delta_s)),
                           _p(_f64(delta_v)), _p(x), _p(_f64(b)), _p(res), _p(sol))
    if rc != 0:
        raise RuntimeError("ho_operator failed rc=%d" % rc)
    return res, sol


def max_threads():
    return lib().ho_max_threads()
