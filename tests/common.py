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


def oracle_params(m1, m2, N, variant, r_f=R_F, rho=RHO, sigma=SIGMA, kappa=KAPPA, eta=ETA, T_=T):
    v = VARIANT[variant] if isinstance(variant, str) else variant
    return O.make_params(m1, m2, N, T_ / N, THETA, R_D, r_f, rho, sigma, kappa, eta, v,
                         DIVS if v in (O.DIV, O.AM_DIV) else None)


def strikes_for(n):
    return [100.0] if n == 1 else [85.0 + 30.0 * k / (n - 1) for k in range(n)]


class OracleSolver:
    """Oracle-backed stand-in with the call signatures of HestonADI.compute_jacobian /
    compute_base_prices -- TESTS ONLY (drives the host-side LM loop where no GPU is available and
    serves as the reference trajectory on the GPU box)."""

    def compute_jacobian(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                         delta_t, num_strikes, grids, U_0, eps=1e-6):
        p = O.make_params(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta)
        return O.jacobian(p, S_0, V_0, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U_0, eps=eps)

    def compute_base_prices(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                            delta_t, num_strikes, grids, ws):
        p = O.make_params(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta)
        return O.base_prices(p, S_0, V_0, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, ws.U)[0]
