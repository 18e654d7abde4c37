"""Shared test inputs: the reference's canonical parameter set (fixtures) and
helpers that build the same seeded batch for the oracle and for libhadi."""
import json
import os

import numpy as np

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
CAN = GOLDEN["canonical"]
S_0, V_0, T = CAN["S_0"], CAN["V_0"], CAN["T"]
R_D, R_F = CAN["r_d"], CAN["r_f"]
RHO, SIGMA, KAPPA, ETA, THETA = CAN["rho"], CAN["sigma"], CAN["kappa"], CAN["eta"], CAN["theta"]
DIVS = (CAN["dividends"]["dates"], CAN["dividends"]["amounts"], CAN["dividends"]["percentages"])
VARIANT = {"EU": O.EU, "AM": O.AM, "DIV": O.DIV, "AM_DIV": O.AM_DIV}


def oracle_grids(m1, m2, strikes, V0=V_0):
    G = [O.grid(m1, 8 * K, S_0, K, K / 5, m2, 5.0, V0, 5.0 / 500) for K in strikes]
    vs, vv, ds, dv = (np.ascontiguousarray(np.stack([g[k] for g in G])) for k in range(4))
    U0 = np.ascontiguousarray(np.stack([np.tile(np.maximum(g[0] - K, 0.0), m2 + 1) for g, K in zip(G, strikes)]))
    return vs, vv, ds, dv, U0


def oracle_params(m1, m2, N, variant, r_f=R_F, rho=RHO, sigma=SIGMA, kappa=KAPPA, eta=ETA, T_=T, option_type=O.CALL,
                  strikes=None):
    v = VARIANT[variant] if isinstance(variant, str) else variant
    return O.make_params(m1, m2, N, T_ / N, THETA, R_D, r_f, rho, sigma, kappa, eta, v,
                         DIVS if v in (O.DIV, O.AM_DIV) else None, option_type=option_type, strikes=strikes)


def put_payoff(vec_s, strikes, m2):
    """U_0 = max(K - s, 0) on every v-row; [n][m]."""
    k = np.asarray(strikes, dtype=np.float64).reshape(-1, 1)
    return np.ascontiguousarray(np.tile(np.maximum(k - vec_s, 0.0), (1, m2 + 1)))


def strikes_for(n):
    return [100.0] if n == 1 else [85.0 + 30.0 * k / (n - 1) for k in range(n)]


class OracleSolver:
    """Oracle-backed stand-in with the call signatures of HestonADI's compute_jacobian* / compute_base_prices*
    launchers -- TESTS ONLY (drives the host-side LM loops where no GPU is available and serves as the
    reference trajectory on the GPU box).  Multi-maturity batches are solved one (N, delta_t) group at a time."""

    @staticmethod
    def _div(dividends):
        return None if dividends is None else (dividends.dates, dividends.amounts, dividends.percentages)

    def _jac(self, variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U_0, eps,
             dividends=None, rows=slice(None)):
        p = O.make_params(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, variant, self._div(dividends))
        return O.jacobian(p, S_0, V_0, grids.Vec_s[rows], grids.Vec_v[rows], grids.Delta_s[rows], grids.Delta_v[rows],
                          np.asarray(U_0)[rows], eps=eps)

    def _base(self, variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U, U_0=None,
              dividends=None, rows=slice(None)):
        p = O.make_params(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, variant, self._div(dividends))
        return O.base_prices(p, S_0, V_0, grids.Vec_s[rows], grids.Vec_v[rows], grids.Delta_s[rows], grids.Delta_v[rows],
                             np.asarray(U)[rows], None if U_0 is None else np.asarray(U_0)[rows])[0]

    def compute_jacobian(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                         delta_t, num_strikes, grids, U_0, eps=1e-6):
        return self._jac(O.EU, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U_0, eps)

    def compute_base_prices(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                            delta_t, num_strikes, grids, ws):
        return self._base(O.EU, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, ws.U)

    def compute_jacobian_american(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                  delta_t, num_strikes, grids, U_0, eps=1e-6):
        return self._jac(O.AM, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U_0, eps)

    def compute_base_prices_american(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                     delta_t, num_strikes, grids, U_0, ws):
        return self._base(O.AM, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, ws.U, U_0)

    def compute_jacobian_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                   delta_t, num_strikes, grids, U_0, dividends, eps=1e-6):
        return self._jac(O.DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U_0, eps,
                         dividends)

    def compute_base_prices_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                      delta_t, num_strikes, grids, ws, dividends):
        return self._base(O.DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, ws.U,
                          dividends=dividends)

    def compute_jacobian_american_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                            theta, delta_t, num_strikes, grids, U_0, dividends, eps=1e-6):
        return self._jac(O.AM_DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, U_0, eps,
                         dividends)

    def compute_base_prices_american_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size,
                                               N, theta, delta_t, num_strikes, grids, U_0, ws, dividends):
        return self._base(O.AM_DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, delta_t, grids, ws.U,
                          U_0, dividends)

    # ---- multi-maturity: group the points by (N, delta_t) ----------------------------------------------
    @staticmethod
    def _groups(points):
        g = {}
        for k, pt in enumerate(points):
            g.setdefault((pt.time_steps, pt.delta_t), []).append(k)
        return [(N, dt, np.array(rows)) for (N, dt), rows in g.items()]

    def _mm_jac(self, variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, U_0, eps,
                dividends=None):
        J, base = np.empty((len(points), 5)), np.empty(len(points))
        for N, dt, rows in self._groups(points):
            J[rows], base[rows] = self._jac(variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, dt,
                                            grids, U_0, eps, dividends, rows)
        return J, base

    def _mm_base(self, variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, U, U_0=None,
                 dividends=None):
        base = np.empty(len(points))
        for N, dt, rows in self._groups(points):
            base[rows] = self._base(variant, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, N, theta, dt, grids, U,
                                    U_0, dividends, rows)
        return base

    def compute_jacobian_multi_maturity(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, theta,
                                        points, n, grids, U_0, eps=1e-6):
        return self._mm_jac(O.EU, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, U_0, eps)

    def compute_base_prices_multi_maturity(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, theta,
                                           points, n, grids, ws):
        return self._mm_base(O.EU, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, ws.U)

    def compute_jacobian_multi_maturity_american_dividends(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                                           total_size, theta, points, n, grids, U_0, dividends, eps=1e-6):
        return self._mm_jac(O.AM_DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, U_0, eps,
                            dividends)

    def compute_base_prices_multi_maturity_american_dividends(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                                              total_size, theta, points, n, grids, U_0, ws, dividends):
        return self._mm_base(O.AM_DIV, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, theta, points, grids, ws.U, U_0,
                             dividends)
