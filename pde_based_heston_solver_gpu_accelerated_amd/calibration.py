"""Levenberg-Marquardt calibration of (kappa, eta, sigma, rho, v0) on top of the batched solver: the host loops of
the reference's calibration drivers (src/heston_calibration.cpp) with the same update, clamps, accept/reject rule
and lambda schedule, made rank-aware: every rank owns a contiguous shard of the calibration instruments, and the
normal equations are all-reduced (31 doubles per iteration, 1 double for the trial error).

    calibrate_european                          test_calibration_european                          :26-512
    calibrate_american                          test_calibration_american                          :515-1033
    calibrate_dividends                         test_calibration_dividends                         :1036-1585
    calibrate_american_dividends                test_calibration_american_dividends                :1588-2160
    calibrate_european_multi_maturity           test_calibration_european_multi_maturity           :2428-2933
    calibrate_american_dividends_multi_maturity test_calibration_american_divident_multi_maturity  :3245-3820

`solver` is anything with the mirrored launchers of solver.HestonADI (`compute_jacobian*`, `compute_base_prices*`).
"""
import collections
import math

import numpy as np

from . import market as _market
from . import solver as _solver
from .distributed import Communicator

RHO_MIN, RHO_MAX = -1.0, 1.0           # heston_calibration.cpp:194
KAPPA_FLOOR, OTHER_FLOOR = 1e-3, 1e-2  # heston_calibration.cpp:286-290
LAMBDA_MIN, LAMBDA_MAX = 1e-7, 1e7     # heston_calibration.cpp:398-408

EU, AM, DIV, AM_DIV = _solver.EU, _solver.AM, _solver.DIV, _solver.AM_DIV

# heston_calibration.cpp:2165-2171
CalibrationPoint = collections.namedtuple("CalibrationPoint", "strike maturity time_steps delta_t global_index")


def make_calibration_points(strikes, maturities, steps_per_year=20, min_steps=20):
    """Flat strike-fastest list, N_m = max(20, int(20 T_m)), dt_m = T_m / N_m (heston_calibration.cpp:2512-2531)."""
    pts = []
    for m, T_m in enumerate(maturities):
        N_m = max(min_steps, int(T_m * steps_per_year))
        dt_m = T_m / N_m
        for s, K in enumerate(strikes):
            pts.append(CalibrationPoint(float(K), float(T_m), N_m, dt_m, m * len(strikes) + s))
    return pts


def clamp_parameters(kappa, eta, sigma, rho, v0):
    return (max(KAPPA_FLOOR, kappa), max(OTHER_FLOOR, eta), max(OTHER_FLOOR, sigma),
            min(RHO_MAX, max(RHO_MIN, rho)), max(OTHER_FLOOR, v0))


class _WS:  # minimal DO_Workspace: the launchers read the initial condition from .U
    pass


def _launchers(solver, variant, S_0, T, r_d, r_f, m1, m2, N, theta, grids, U_0, dividends, points):
    """Two closures (params -> (J, base)), (params -> trial prices) over the launcher pair of the variant."""
    n_loc = grids.Vec_s.shape[0]
    total_size = (m1 + 1) * (m2 + 1)
    if variant in (DIV, AM_DIV) and dividends is None:
        raise ValueError("this variant needs a dividend schedule")

    def workspace():
        ws = _WS()
        ws.U = _clone(U_0)                                               # deep_copy(workspace.U, U_0)
        return ws

    if points is not None:                                               # multi-maturity launchers
        if variant == EU:
            def jac(k, e, s, r, v, eps):
                return solver.compute_jacobian_multi_maturity(S_0, v, r_d, r_f, r, s, k, e, m1, m2, total_size, theta,
                                                              points, n_loc, grids, U_0, eps=eps)

            def base(k, e, s, r, v):
                return solver.compute_base_prices_multi_maturity(S_0, v, r_d, r_f, r, s, k, e, m1, m2, total_size,
                                                                 theta, points, n_loc, grids, workspace())
        elif variant == AM_DIV:
            def jac(k, e, s, r, v, eps):
                return solver.compute_jacobian_multi_maturity_american_dividends(
                    S_0, v, r_d, r_f, r, s, k, e, m1, m2, total_size, theta, points, n_loc, grids, U_0, dividends,
                    eps=eps)

            def base(k, e, s, r, v):
                return solver.compute_base_prices_multi_maturity_american_dividends(
                    S_0, v, r_d, r_f, r, s, k, e, m1, m2, total_size, theta, points, n_loc, grids, U_0, workspace(),
                    dividends)
        else:
            raise ValueError("the reference has multi-maturity launchers for EU and AM_DIV only")
        return jac, base

    delta_t = T / N
    head = lambda k, e, s, r, v: (S_0, v, T, r_d, r_f, r, s, k, e, m1, m2, total_size, N, theta, delta_t, n_loc, grids)
    if variant == EU:
        jac = lambda k, e, s, r, v, eps: solver.compute_jacobian(*head(k, e, s, r, v), U_0, eps=eps)
        base = lambda k, e, s, r, v: solver.compute_base_prices(*head(k, e, s, r, v), workspace())
    elif variant == AM:
        jac = lambda k, e, s, r, v, eps: solver.compute_jacobian_american(*head(k, e, s, r, v), U_0, eps=eps)
        base = lambda k, e, s, r, v: solver.compute_base_prices_american(*head(k, e, s, r, v), U_0, workspace())
    elif variant == DIV:
        jac = lambda k, e, s, r, v, eps: solver.compute_jacobian_dividends(*head(k, e, s, r, v), U_0, dividends, eps=eps)
        base = lambda k, e, s, r, v: solver.compute_base_prices_dividends(*head(k, e, s, r, v), workspace(), dividends)
    elif variant == AM_DIV:
        jac = lambda k, e, s, r, v, eps: solver.compute_jacobian_american_dividends(*head(k, e, s, r, v), U_0, dividends,
                                                                                    eps=eps)
        base = lambda k, e, s, r, v: solver.compute_base_prices_american_dividends(*head(k, e, s, r, v), U_0, workspace(),
                                                                                   dividends)
    else:
        raise ValueError("unknown variant %r" % (variant,))
    return jac, base


def calibrate(solver, variant, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
              market_prices, dividends=None, calibration_points=None, max_iter=20, tol=0.1, delta_tol=None, eps=1e-6,
              lam=0.01, comm=None, lm_partials=_solver.lm_partials, lm_solve=_solver.lm_solve):
    """The LM loop shared by all drivers (heston_calibration.cpp:204-417).  `grids`, `U_0`, `market_prices` (and
    `calibration_points` for the multi-maturity drivers, which then ignore T and N) are this rank's shard.
    Stops when ||delta||_2 < delta_tol (default: tol) or sum r^2 < tol.  Returns a dict with the calibrated
    parameters, final error, iteration count, PDE-solve count, the last computed model prices and the trajectory."""
    comm = comm or Communicator()
    n_loc = grids.Vec_s.shape[0]
    delta_tol = tol if delta_tol is None else delta_tol
    market = np.asarray(market_prices, dtype=np.float64)
    jac, base_fn = _launchers(solver, variant, S_0, T, r_d, r_f, m1, m2, N, theta, grids, U_0, dividends,
                              calibration_points)
    cur = (kappa, eta, sigma, rho, V_0)
    final_error, iteration_count, converged = 100.0, 0, False
    history = []
    model = None

    # Device-resident shards (CUDA tensors through a real handle): the Jacobian rows and prices stay in HBM and
    # J^T J / J^T r / sum r^2 are reduced there (hadi_lm_partials_device) -- 31 doubles per iteration leave the GPU, as
    # after the reference's KokkosBlas gemm/gemv (jacobian_computation.cpp:117,154).  Host arrays take the host loop.
    on_device = _solver._is_device(U_0) and hasattr(solver, "_h") and lm_partials is _solver.lm_partials
    market_d = None
    if on_device:
        import torch
        market_d = torch.from_numpy(np.ascontiguousarray(market)).to(U_0.device)

    for it in range(max_iter):
        J, base = jac(*cur, eps)
        model = base
        if on_device:
            part_loc = _solver.lm_partials_device(solver, J, base, market_d)
        else:
            J, base = _to_numpy(J), _to_numpy(base)
            model = base
            resid = market - base                                        # heston_calibration.cpp:271-275
            part_loc = lm_partials(J, resid)
        part = comm.allreduce_sum(part_loc)                              # the only collective of the step
        delta = lm_solve(part, lam)
        new = clamp_parameters(*(c + d for c, d in zip(cur, delta)))
        delta_norm = float(np.sqrt(np.sum(delta * delta)))
        current_error = float(part[30])
        history.append({"iter": it + 1, "params": cur, "error": current_error, "lambda": lam,
                        "delta": delta.copy(), "trial": new})
        if delta_norm < delta_tol or current_error < tol:                # heston_calibration.cpp:322-338
            converged = True
            cur = new
            final_error = current_error
            iteration_count = it + 1
            break
        trial = base_fn(*new)
        model = trial
        if on_device:
            solver.wait_stream()  # (trial is complete; torch ops below run on torch's stream)
            loc_err = float(((market_d - trial) ** 2).sum().item())      # one double leaves the GPU
        else:
            trial = _to_numpy(trial)
            model = trial
            r_new = market - trial
            loc_err = float(np.sum(r_new * r_new))
        new_error = float(comm.allreduce_sum(np.array([loc_err]))[0])
        history[-1]["trial_error"] = new_error
        if new_error < current_error:                                    # heston_calibration.cpp:398-408
            cur = new
            lam = max(lam / 10.0, LAMBDA_MIN)
        else:
            lam = min(lam * 10.0, LAMBDA_MAX)
        final_error = min(new_error, current_error)
        iteration_count = it + 1
    n_glob = int(round(comm.allreduce_sum(np.array([float(n_loc)]))[0]))
    return {"kappa": cur[0], "eta": cur[1], "sigma": cur[2], "rho": cur[3], "v0": cur[4],
            "final_error": final_error, "iterations": iteration_count, "converged": converged,
            "pde_solves": n_glob * 7 * iteration_count - n_glob,          # heston_calibration.cpp:428
            "initial": (kappa, eta, sigma, rho, V_0), "model_prices": None if model is None else _to_numpy(model),
            "history": history}


def calibrate_european(solver, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                       market_prices, max_iter=15, tol=0.1, **kw):
    return calibrate(solver, EU, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                     market_prices, max_iter=max_iter, tol=tol, **kw)


def calibrate_american(solver, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                       market_prices, max_iter=15, tol=0.1, **kw):
    return calibrate(solver, AM, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                     market_prices, max_iter=max_iter, tol=tol, **kw)


def calibrate_dividends(solver, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                        market_prices, dividends, max_iter=20, tol=0.1, **kw):
    return calibrate(solver, DIV, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                     market_prices, dividends=dividends, max_iter=max_iter, tol=tol, **kw)


def calibrate_american_dividends(solver, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                                 market_prices, dividends, max_iter=20, tol=0.1, **kw):
    return calibrate(solver, AM_DIV, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                     market_prices, dividends=dividends, max_iter=max_iter, tol=tol, **kw)


def multi_maturity_tolerances(n_points):
    """tol = 0.1 sqrt(n), delta_tol = 0.1 (1 + ln n): heston_calibration.cpp:2544-2545."""
    return 1e-1 * math.sqrt(n_points), 1e-1 * (1.0 + math.log(n_points))


def calibrate_european_multi_maturity(solver, S_0, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, theta,
                                      calibration_points, grids, U_0, market_prices, max_iter=15, tol=None,
                                      delta_tol=None, n_total=None, **kw):
    """`n_total` = global number of calibration points (defaults to this rank's, i.e. single rank)."""
    t, dtol = multi_maturity_tolerances(n_total or len(calibration_points))
    return calibrate(solver, EU, S_0, None, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, None, theta, grids, U_0,
                     market_prices, calibration_points=calibration_points, max_iter=max_iter,
                     tol=t if tol is None else tol, delta_tol=dtol if delta_tol is None else delta_tol, **kw)


def calibrate_american_dividends_multi_maturity(solver, S_0, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, theta,
                                                calibration_points, grids, U_0, market_prices, dividends, max_iter=20,
                                                tol=0.1, **kw):
    return calibrate(solver, AM_DIV, S_0, None, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, None, theta, grids, U_0,
                     market_prices, dividends=dividends, calibration_points=calibration_points, max_iter=max_iter,
                     tol=tol, **kw)


def export_calibration_csv(path, result, S_0, r_d, strikes, market_prices, T=None, maturities=None, dividends=None,
                           elapsed_s=0.0, epsilon=0.01):
    """The drivers' result files (`fitted_heston_vs_market*.csv`): a '#' metadata line, then per instrument the
    market and fitted price with implied-vol difference (heston_calibration.cpp:443-511; multi-maturity layout
    with both implied vols :2851-2931; dividend drivers invert on the dividend-adjusted spot :1498-1530).
    `strikes` / `market_prices` cover ALL instruments in global order, `result["model_prices"]` likewise."""
    fitted = np.asarray(result["model_prices"], dtype=np.float64)
    market = np.asarray(market_prices, dtype=np.float64)
    k0, e0, s0, r0, v0 = result["initial"]
    tail = (", init_kappa=%g, init_eta=%g, init_sigma=%g, init_rho=%g, init_v0=%g, kappa=%g, eta=%g, sigma=%g, rho=%g, "
            "v0=%g\n" % (k0, e0, s0, r0, v0, result["kappa"], result["eta"], result["sigma"], result["rho"],
                          result["v0"]))

    def spot(T_):
        if dividends is None:
            return S_0
        S_adj = S_0
        for date, amount, pct in zip(dividends.dates, dividends.amounts, dividends.percentages):
            if date < T_:
                S_adj -= amount * math.exp(-r_d * date)
                S_adj -= (S_0 * pct) * math.exp(-r_d * date)
        return S_adj

    with open(path, "w") as out:
        if maturities is None:
            out.write("# %d options, Time=%g s, FinalError=%g, iterationCount=%d, TotalPdeSolves=%d" %
                      (len(strikes), elapsed_s, result["final_error"], result["iterations"], result["pde_solves"]) + tail)
            out.write("Strike,MarketPrice,FittedPrice,IVDifference\n")
            S = spot(T)
            for K, mp, fp in zip(strikes, market, fitted):
                d = abs(_market.reverse_BS(S, K, r_d, T, 0.5, mp, epsilon) - _market.reverse_BS(S, K, r_d, T, 0.5, fp, epsilon))
                out.write("%g,%g,%g,%g\n" % (K, mp, fp, d))
        else:
            out.write("# Calibration with %d maturities, %d strikes per maturity, Time=%g s, FinalError=%g, "
                      "IterationCount=%d, TotalPdeSolves=%d" % (len(maturities), len(strikes), elapsed_s,
                                                                result["final_error"], result["iterations"],
                                                                result["pde_solves"]) + tail)
            out.write("Maturity,Strike,MarketPrice,FittedPrice,MarketIV,FittedIV,IVDifference\n")
            for m, T_m in enumerate(maturities):
                S = spot(T_m)
                for s, K in enumerate(strikes):
                    idx = m * len(strikes) + s
                    miv = _market.reverse_BS(S, K, r_d, T_m, 0.5, market[idx], epsilon)
                    fiv = _market.reverse_BS(S, K, r_d, T_m, 0.5, fitted[idx], epsilon)
                    out.write("%g,%g,%g,%g,%g,%g,%g\n" % (T_m, K, market[idx], fitted[idx], miv, fiv, abs(miv - fiv)))


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _clone(x):
    return x.clone() if hasattr(x, "clone") else np.array(x, copy=True)
