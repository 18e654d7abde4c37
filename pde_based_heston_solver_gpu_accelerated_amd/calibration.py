"""Levenberg-Marquardt calibration of (kappa, eta, sigma, rho, v0) on top of the batched solver: the
host loop of the reference's `test_calibration_european` (src/heston_calibration.cpp:204-417) with the
same update, clamps, accept/reject rule and lambda schedule, made rank-aware: every rank owns a
contiguous shard of the strikes, and the normal equations are all-reduced (31 doubles per iteration,
1 double for the trial error).

`solver` is anything with `compute_jacobian(...)` and `compute_base_prices(...)` in the mirrored
signatures of solver.HestonADI.
"""
import numpy as np

from . import solver as _solver
from .distributed import Communicator

RHO_MIN, RHO_MAX = -1.0, 1.0           # heston_calibration.cpp:194
KAPPA_FLOOR, OTHER_FLOOR = 1e-3, 1e-2  # heston_calibration.cpp:286-290
LAMBDA_MIN, LAMBDA_MAX = 1e-7, 1e7     # heston_calibration.cpp:398-408


def clamp_parameters(kappa, eta, sigma, rho, v0):
    return (max(KAPPA_FLOOR, kappa), max(OTHER_FLOOR, eta), max(OTHER_FLOOR, sigma),
            min(RHO_MAX, max(RHO_MIN, rho)), max(OTHER_FLOOR, v0))


def calibrate_european(solver, S_0, T, r_d, r_f, kappa, eta, sigma, rho, V_0, m1, m2, N, theta, grids, U_0,
                       market_prices, max_iter=20, tol=0.1, eps=1e-6, lam=0.01, comm=None,
                       lm_partials=_solver.lm_partials, lm_solve=_solver.lm_solve):
    """Returns a dict with the calibrated parameters, final error, iteration count and the trajectory.
    `grids`, `U_0`, `market_prices` are this rank's shard."""
    comm = comm or Communicator()
    n_loc = grids.Vec_s.shape[0]
    total_size = (m1 + 1) * (m2 + 1)
    delta_t = T / N
    market = np.asarray(market_prices, dtype=np.float64)
    cur = (kappa, eta, sigma, rho, V_0)
    final_error, iteration_count, converged = 100.0, 0, False
    history = []

    class _WS:  # minimal DO_Workspace: the launchers read the initial condition from .U
        pass

    for it in range(max_iter):
        ck, ce, cs, cr, cv = cur
        J, base = solver.compute_jacobian(S_0, cv, T, r_d, r_f, cr, cs, ck, ce, m1, m2, total_size, N, theta,
                                          delta_t, n_loc, grids, U_0, eps=eps)
        J, base = _to_numpy(J), _to_numpy(base)
        resid = market - base                                            # heston_calibration.cpp:271-275
        part = comm.allreduce_sum(lm_partials(J, resid))                 # the only collective of the step
        delta = lm_solve(part, lam)
        new = clamp_parameters(ck + delta[0], ce + delta[1], cs + delta[2], cr + delta[3], cv + delta[4])
        delta_norm = float(np.sqrt(np.sum(delta * delta)))
        current_error = float(part[30])
        history.append({"iter": it + 1, "params": cur, "error": current_error, "lambda": lam,
                        "delta": delta.copy(), "trial": new})
        if delta_norm < tol or current_error < tol:                      # heston_calibration.cpp:322-338
            converged = True
            cur = new
            final_error = current_error
            iteration_count = it + 1
            break
        ws = _WS()
        ws.U = _clone(U_0)                                               # deep_copy(workspace.U, U_0)
        nk, ne, ns, nr, nv = new
        trial = solver.compute_base_prices(S_0, nv, T, r_d, r_f, nr, ns, nk, ne, m1, m2, total_size, N, theta,
                                           delta_t, n_loc, grids, ws)
        r_new = market - _to_numpy(trial)
        new_error = float(comm.allreduce_sum(np.array([np.sum(r_new * r_new)]))[0])
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
            "history": history}


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _clone(x):
    return x.clone() if hasattr(x, "clone") else np.array(x, copy=True)
