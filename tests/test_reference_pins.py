"""CPU: the oracle against the data the REFERENCE ITSELF holds for this path (as opposed to the survey-recorded outputs
in tests/golden/reference_known_answers.json, whose generating build cannot be repeated here):

  * src/solver.cpp:1653-1692 `test_convergence`: target 8.8948693600540167, sweep m2 in {15,25,50,75,100,125,150},
    m1 = 2 m2, N = 20, theta = 0.8 -- the relative errors it exports must fall monotonically, and the target itself is
    the semi-analytic Heston price (reproduced below by quadrature of the characteristic function: an anchor that does
    not depend on any build of the reference);
  * the hard-coded print targets src/device_solver.cpp:778 (8.8943383103218502), src/solver.cpp:834 (3.839290124997349),
    src/solver.cpp:1091 (5.285130942409008) with their documented gaps;
  * the per-operator acceptance drivers (src/hes_a0_kernels.cpp:8-121, src/hes_a1_kernels.cpp:130-277,
    src/hes_a2_shuffled_kernels.cpp:13-157): x == 1, b == 2, printed residual ||x - theta dt A x - b||_2.
The GPU twins of these checks live in tests/test_gpu_pins.py."""
import numpy as np
import pytest
from scipy.integrate import quad

from oracle import oracle as O

import common as Cm

REF_CONVERGENCE_TARGET = 8.8948693600540167   # src/solver.cpp:1666
REF_PRINT_TARGET_EU = 8.8943383103218502      # src/device_solver.cpp:778, src/solver.cpp:399,502,602
REF_PRINT_TARGET_DIV = 3.839290124997349      # src/solver.cpp:834
REF_PRINT_TARGET_AMDIV = 5.285130942409008    # src/solver.cpp:1091
CONVERGENCE_M2 = (15, 25, 50, 75, 100, 125, 150)  # src/solver.cpp:1669


def heston_call_semi_analytic(S0, K, T, r_d, r_f, kappa, eta, sigma, rho, v0):
    """Heston (1993) call by Gil-Pelaez inversion of the characteristic function ("little trap" form)."""
    def cf(u):
        x = np.log(S0) + (r_d - r_f) * T
        d = np.sqrt((rho * sigma * 1j * u - kappa) ** 2 + sigma ** 2 * (1j * u + u * u))
        g = (kappa - rho * sigma * 1j * u - d) / (kappa - rho * sigma * 1j * u + d)
        C = kappa * eta / sigma ** 2 * ((kappa - rho * sigma * 1j * u - d) * T - 2 * np.log((1 - g * np.exp(-d * T)) / (1 - g)))
        D = (kappa - rho * sigma * 1j * u - d) / sigma ** 2 * (1 - np.exp(-d * T)) / (1 - g * np.exp(-d * T))
        return np.exp(1j * u * x + C + D * v0)
    k = np.log(K)
    kw = dict(limit=2000, epsabs=1e-13, epsrel=1e-13)
    P1 = 0.5 + quad(lambda u: (np.exp(-1j * u * k) * cf(u - 1j) / (1j * u * cf(-1j))).real, 1e-12, 400, **kw)[0] / np.pi
    P2 = 0.5 + quad(lambda u: (np.exp(-1j * u * k) * cf(u) / (1j * u)).real, 1e-12, 400, **kw)[0] / np.pi
    return S0 * np.exp(-r_f * T) * P1 - K * np.exp(-r_d * T) * P2


def oracle_price(m1, m2, N, K=100.0, variant="EU", theta=Cm.THETA):
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [K])
    p = Cm.oracle_params(m1, m2, N, variant)
    p.theta = theta
    U, _, _ = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    return U[O.find_s_index(vs[0], Cm.S_0) + O.find_v_index(vv[0], Cm.V_0) * (m1 + 1)]


def test_convergence_target_is_the_semi_analytic_heston_price():
    """The reference's ref_price (solver.cpp:1666) to all the digits a 1e-13 quadrature gives."""
    p = heston_call_semi_analytic(Cm.S_0, 100.0, Cm.T, Cm.R_D, Cm.R_F, Cm.KAPPA, Cm.ETA, Cm.SIGMA, Cm.RHO, Cm.V_0)
    assert abs(p - REF_CONVERGENCE_TARGET) < 5e-11


def test_reference_convergence_sweep():
    """test_convergence (solver.cpp:1653-1692) on the oracle: relative error against the reference's target falls
    monotonically over the sweep and ends at the N = 20 time-discretisation floor."""
    errs = [abs(oracle_price(2 * m2, m2, 20) - REF_CONVERGENCE_TARGET) / REF_CONVERGENCE_TARGET for m2 in CONVERGENCE_M2]
    assert all(a > b for a, b in zip(errs, errs[1:])), errs
    assert 8e-3 < errs[0] < 9e-3 and errs[-1] < 2.3e-3, errs
    # the floor is the time step (theta = 0.8 Douglas is first order): refining N on the 200x100 grid halves the error
    e = [abs(oracle_price(200, 100, N) - REF_CONVERGENCE_TARGET) for N in (20, 40, 80, 160)]
    assert all(0.45 < b / a < 0.65 for a, b in zip(e, e[1:])), e
    # and Richardson extrapolation in dt on that grid lands within 2e-4 of the target (spatial error of 200x100)
    p80, p160 = oracle_price(200, 100, 80), oracle_price(200, 100, 160)
    assert abs(2 * p160 - p80 - REF_CONVERGENCE_TARGET) < 2e-4 * REF_CONVERGENCE_TARGET


def test_reference_print_targets_and_their_documented_gaps():
    """The three hard-coded print targets are external ("from Python/Monte Carlo"), not regression values:
    8.8943383 sits 5.3e-4 below the semi-analytic price, the scheme's own output on the printed 50x25x20 grid is 4.3e-2
    away and closes in under refinement; 3.8392901 is approached by the dividend scheme the same way; 5.2851309 is printed
    by the K = 95 American-dividend test although it belongs to K = 100 (the K = 95 price is 8.51)."""
    assert 5.2e-4 < REF_CONVERGENCE_TARGET - REF_PRINT_TARGET_EU < 5.4e-4
    coarse, fine = oracle_price(50, 25, 20), oracle_price(200, 100, 160)
    assert abs(coarse - 8.8512320311290900) < 1e-12              # survey-recorded reference output, same grid
    assert 4.2e-2 < REF_PRINT_TARGET_EU - coarse < 4.4e-2 and abs(fine - REF_PRINT_TARGET_EU) < 4e-3
    dc, df = oracle_price(50, 25, 20, variant="DIV"), oracle_price(200, 100, 160, variant="DIV")
    assert abs(dc - REF_PRINT_TARGET_DIV) < 1.3e-2 and abs(df - REF_PRINT_TARGET_DIV) < abs(dc - REF_PRINT_TARGET_DIV)
    a95 = oracle_price(50, 25, 20, K=95.0, variant="AM_DIV")
    a100c, a100f = oracle_price(50, 25, 20, variant="AM_DIV"), oracle_price(200, 100, 160, variant="AM_DIV")
    assert abs(a95 - 8.5105730742666701) < 1e-12 and abs(a95 - REF_PRINT_TARGET_AMDIV) > 3.0
    assert abs(a100f - REF_PRINT_TARGET_AMDIV) < abs(a100c - REF_PRINT_TARGET_AMDIV) < 0.15


def reference_operator_setup(m1, m2):
    """create_test_grid (grid.cpp:99-110) and the drivers' constants (hes_a1_kernels.cpp:138-147,
    hes_a2_shuffled_kernels.cpp:17-27)."""
    vs, vv, ds, dv = O.grid(m1, 800.0, 100.0, 100.0, 20.0, m2, 5.0, 0.04, 0.01)
    p = O.make_params(m1, m2, 1, 1.0 / 40.0, 0.8, 0.025, 0.0, Cm.RHO, 0.3, 1.5, 0.04, O.EU)
    return p, vs, vv, ds, dv


@pytest.mark.parametrize("which,m1,m2", [(1, 150, 75), (2, 100, 75)])
def test_reference_operator_acceptance_checks(which, m1, m2):
    """debugging_test_device_adi_multiple_instances (A1, 150x75) and test_device_a2_shuffled_multiple_instances (A2,
    100x75): x == 1, b == 2; solve (I - theta dt A) x = b; printed residual ||x - theta dt A x - b||_2 must be round-off."""
    p, vs, vv, ds, dv = reference_operator_setup(m1, m2)
    m = (m1 + 1) * (m2 + 1)
    x, b = np.ones(m), 2.0 * np.ones(m)
    _, sol = O.operator(p, which, vs, vv, ds, dv, x, b)
    Ax, _ = O.operator(p, which, vs, vv, ds, dv, sol)
    res = sol - p.theta * p.delta_t * Ax - b
    assert np.sqrt((res * res).sum()) < 1e-11
    assert np.abs(sol - 2.0).max() > 1e-6  # (the solve did something)
    # A0 driver (hes_a0_kernels.cpp:8-121): the mixed-derivative stencil annihilates constants
    A0x, _ = O.operator(p, 0, vs, vv, ds, dv, x)
    assert np.abs(A0x).max() < 1e-9


def test_put_boundary_data_on_the_oracle():
    """HADI_PUT is not a reference feature; its checker (the oracle's option_type = 1) is validated here on its own:
    put-call parity against the call path -- exact up to the discrete discount factor of the Douglas step, 1.8e-4 from the
    continuous forward at N = 100 -- the American put dominating the European one and its payoff, and the price
    approaching the semi-analytic Heston put."""
    m1, m2, N, K = 100, 50, 100, 100.0
    vs, vv, ds, dv, C0 = Cm.oracle_grids(m1, m2, [K])
    P0 = Cm.put_payoff(vs, [K], m2)
    pc = Cm.oracle_params(m1, m2, N, "EU")
    pp = Cm.oracle_params(m1, m2, N, "EU", option_type=O.PUT, strikes=[K])
    Uc, _, _ = O.solve(pc, vs[0], vv[0], ds[0], dv[0], C0[0])
    Up, _, _ = O.solve(pp, vs[0], vv[0], ds[0], dv[0], P0[0])
    node = O.find_s_index(vs[0], Cm.S_0) + O.find_v_index(vv[0], Cm.V_0) * (m1 + 1)
    th, dt, r = Cm.THETA, Cm.T / N, Cm.R_D
    g = ((1 - dt * r + th * dt * 0.5 * r) / (1 + th * dt * 0.5 * r) + th * dt * 0.5 * r) / (1 + th * dt * 0.5 * r)
    assert abs((Uc[node] - Up[node]) - (Cm.S_0 - K * g ** N)) < 1e-7          # discrete parity
    assert abs((Uc[node] - Up[node]) - (Cm.S_0 - K * np.exp(-r * Cm.T))) < 2e-4  # continuous forward: O(dt)
    D = (Uc - Up).reshape(m2 + 1, m1 + 1)[:20, :60]                            # region around the price node
    assert np.abs(D - (vs[0][None, :60] - K * g ** N)).max() < 1e-5
    pa = Cm.oracle_params(m1, m2, N, "AM", option_type=O.PUT, strikes=[K])
    Ua, lam, _ = O.solve(pa, vs[0], vv[0], ds[0], dv[0], P0[0], P0[0])
    inner = np.ones((m2 + 1, m1 + 1), bool)
    inner[:, m1] = False  # (the s_max column, where lambda_bar is held at 0, carries a 5e-3 wiggle at v > 3)
    assert (Ua >= P0[0] - 1e-12).all() and (Ua >= Up - 1e-9)[inner.ravel()].all() and Ua[node] > Up[node] + 1e-3 and lam.max() > 0
    put_exact = heston_call_semi_analytic(Cm.S_0, K, Cm.T, r, 0.0, Cm.KAPPA, Cm.ETA, Cm.SIGMA, Cm.RHO, Cm.V_0) - Cm.S_0 + K * np.exp(-r * Cm.T)
    assert abs(Up[node] - put_exact) < 1.2e-2
    # the s = 0 column decays with the discrete discount factor on every v-row, the last one included (there A2 is empty
    # and b2 supplies the missing half of the reaction term -- the far-field value K e^{-r_d t} is exact on that row)
    F = Up.reshape(m2 + 1, m1 + 1)
    assert abs(F[3, 0] - K * g ** N) < 1e-9 and abs(F[m2, 0] - K * np.exp(-r * Cm.T)) < 2e-4
