"""The oracle against every reference output recorded for this path (SURVEY.md section 8(c),
tests/golden/reference_known_answers.json).  CPU only."""
import numpy as np
import pytest

import pde_based_heston_solver_gpu_accelerated_amd as H
from oracle import oracle as O

import common as Cm


def _price(case):
    m1, m2, N, K = case["m1"], case["m2"], case["N"], float(case["K"])
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [K])
    p = Cm.oracle_params(m1, m2, N, case["variant"])
    U, _, _ = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    i_s, i_v = O.find_s_index(vs[0], Cm.S_0), O.find_v_index(vv[0], Cm.V_0)
    return U[i_s + i_v * (m1 + 1)], i_s, i_v


CASES = [c for c in Cm.GOLDEN["prices"] if not c.get("slow")]
SLOW = [c for c in Cm.GOLDEN["prices"] if c.get("slow")]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%dx%d_K%s" % (c["variant"], c["m1"], c["m2"], c["N"], c["K"]))
def test_price_matches_reference_output(case):
    price, i_s, i_v = _price(case)
    # 17-digit records must match to the last bit; shorter prints to their printed precision
    tol = 0.0 if case["digits"] >= 17 else 0.5 * 10.0 ** (2 - case["digits"])
    assert abs(price - case["price"]) <= tol, (price, case["price"])
    if "idx_s" in case:
        assert (i_s, i_v) == (case["idx_s"], case["idx_v"])


@pytest.mark.slow
@pytest.mark.parametrize("case", SLOW, ids=lambda c: "%s_%dx%dx%d" % (c["variant"], c["m1"], c["m2"], c["N"]))
def test_price_headline_config(case):
    """BASELINE config 2 (512x256, 1000 steps): ~8 s of CPU."""
    price, i_s, i_v = _price(case)
    assert price == case["price"]
    assert (i_s, i_v) == (case["idx_s"], case["idx_v"])


def test_jacobian_row_matches_reference_output():
    j = Cm.GOLDEN["jacobian"]
    m1, m2, N = j["m1"], j["m2"], j["N"]
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [float(k) for k in j["strikes"]])
    p = Cm.oracle_params(m1, m2, N, "EU")
    J, base = O.jacobian(p, Cm.S_0, Cm.V_0, vs, vv, ds, dv, U0, eps=j["eps"])
    assert abs(base[0] - j["base_price_0"]) < 1e-14  # printed to 14 decimals
    assert np.abs(J[0] - np.array(j["J_row_0"])).max() < 0.5e-12  # printed to 12 decimals


def test_craig_sneyd_price_matches_reference_output():
    """CS_scheme_shuffled (solver.hpp:781-907) on the reference's 50x25x20 test grid: all 16 recorded digits."""
    c = Cm.GOLDEN["craig_sneyd"]
    m1, m2, N, K = c["m1"], c["m2"], c["N"], float(c["K"])
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [K])
    p = Cm.oracle_params(m1, m2, N, "EU")
    p.scheme = 1
    U, _, _ = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0])
    price = U[O.find_s_index(vs[0], Cm.S_0) + O.find_v_index(vv[0], Cm.V_0) * (m1 + 1)]
    assert abs(price - c["price"]) <= 1e-15 * c["price"]
    pa = Cm.oracle_params(m1, m2, N, "AM")
    pa.scheme = 1
    with pytest.raises(RuntimeError):  # the reference has CS for European options only
        O.solve(pa, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])


def test_external_targets_are_loose():
    """The prices hard-coded in the reference's prints are external targets ~1e-3 away from the
    scheme's own 50x25x20 output -- documents why they cannot pin parity."""
    price, _, _ = _price({"variant": "EU", "m1": 50, "m2": 25, "N": 20, "K": 100})
    assert 1e-4 < abs(price - Cm.GOLDEN["external_targets"]["eu_call_K100"]) < 0.1


def test_implicit_solve_residuals():
    """The reference's own acceptance check is the printed residual ||x - theta dt A x - b||
    (hes_a1_kernels.cpp:262-276, hes_a2_shuffled_kernels.cpp:142-156): verify it on the oracle's
    step-1 intermediates, rebuilding A1/A2 products by finite application of the solvers."""
    m1, m2, N = 40, 20, 10
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [100.0])
    p = Cm.oracle_params(m1, m2, N, "EU")
    _, _, d = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0], dump_step=1)
    # (I - theta dt A1) Y1 = Y0rhs  <=>  A1 Y1 = (Y1 - Y0rhs)/(theta dt); check by re-applying the
    # explicit operator through a second oracle call that starts from Y1 and dumps A1U.
    _, _, d2 = O.solve(p, vs[0], vv[0], ds[0], dv[0], d["Y1"], d["Y1"], dump_step=1)
    thdt = Cm.THETA * Cm.T / N
    res = d["Y1"] - thdt * d2["A1U"] - d["Y0rhs"]
    assert np.abs(res).max() < 1e-10 * max(1.0, np.abs(d["Y0rhs"]).max())
    _, _, d3 = O.solve(p, vs[0], vv[0], ds[0], dv[0], d["Unext"], d["Unext"], dump_step=1)
    res2 = d["Unext"] - thdt * d3["A2U"] - d["Y1rhs"]
    assert np.abs(res2).max() < 1e-10 * max(1.0, np.abs(d["Y1rhs"]).max())


def _reference_calibration_setup():
    """test_calibration_european's setup, heston_calibration.cpp:26-120."""
    g = Cm.GOLDEN["calibration_european"]
    strikes = [Cm.S_0 * 0.7 + i for i in range(60)]
    m1, m2, N = 50, 25, 20
    grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, strikes)
    U0 = grids.call_payoff(strikes)
    market = H.market.generate_market_data(Cm.S_0, Cm.T, Cm.R_D, strikes)
    args = (Cm.S_0, Cm.T, Cm.R_D, Cm.R_F, Cm.KAPPA, Cm.ETA, Cm.SIGMA, Cm.RHO, Cm.V_0, m1, m2, N, Cm.THETA, grids, U0, market)
    return g, args


def check_calibration_against_record(res, g, tol=None, kappa_tol=None):
    if tol is None:
        tol = 0.5 * 10.0 ** (1 - g["digits"])  # the record prints 6 significant digits
    if kappa_tol is None:
        kappa_tol = tol
    assert res["converged"] and res["iterations"] == g["iterations"] and res["pde_solves"] == g["pde_solves"]
    assert abs(res["final_error"] - g["final_error"]) <= tol * g["final_error"]
    for k in ("kappa", "eta", "sigma", "rho", "v0"):
        assert abs(res[k] - g[k]) <= (kappa_tol if k == "kappa" else tol) * abs(g[k]), (k, res[k], g[k])


def test_full_lm_calibration_reproduces_reference_run():
    """N1 + N4 pin: oracle solves + the LM host loop + Black-Scholes market data reproduce the recorded output of
    the reference's test_calibration_european: 4 iterations, 1620 PDE solves, final error and all
    five calibrated parameters to the 6 printed digits."""
    g, args = _reference_calibration_setup()
    res = H.calibrate_european(Cm.OracleSolver(), *args, max_iter=15, tol=0.1)
    check_calibration_against_record(res, g)
