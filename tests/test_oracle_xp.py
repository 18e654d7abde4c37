"""The extended-precision adjudicator (oracle/heston_oracle_xp.c: the oracle's source compiled with binary128 arithmetic)
against the pinned fp64 oracle: same scheme, same quirks, same dividend dating -- only the round-off differs."""
import numpy as np
import pytest

from oracle import oracle as O

import common as Cm


@pytest.mark.parametrize("variant,m1,m2,N", [("EU", 50, 25, 20), ("AM", 50, 25, 20), ("DIV", 50, 25, 20), ("AM_DIV", 50, 25, 20),
                                             ("EU", 100, 50, 200)])
def test_adjudicator_reproduces_the_pinned_oracle_to_round_off(variant, m1, m2, N):
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [100.0])
    p = Cm.oracle_params(m1, m2, N, variant)
    U, lam, _ = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    Ux, lx = O.solve_xp(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    assert np.abs(U - Ux).max() < 5e-13 * np.abs(Ux).max()
    if lam is not None:
        assert np.abs(lam - lx).max() < 1e-9 * max(1.0, np.abs(lx).max())
    # the recorded reference price of this case sits on the exact-arithmetic result to round-off as well
    key = {"EU": "eu", "AM": "am", "DIV": "div", "AM_DIV": "amdiv"}[variant]
    rec = [r for r in Cm.GOLDEN["prices"] if r.get("m1") == m1 and r.get("m2") == m2 and r.get("N") == N and
           r.get("variant", "").lower().replace("_", "") == key and r.get("K", 100) == 100]
    if rec:
        is_, iv = O.find_s_index(vs[0], Cm.S_0), O.find_v_index(vv[0], Cm.V_0)
        assert abs(Ux[is_ + iv * (m1 + 1)] - rec[0]["price"]) < 1e-11


def test_adjudicator_dates_dividends_in_fp64():
    """12 * 0.05 = 0.6000000000000001 in fp64 decides the step the 0.6 dividend lands on (device_solver.hpp:426-447); the
    adjudicator must take the same step, or it would adjudicate a different problem."""
    m1, m2, N = 30, 12, 20  # dt = 0.05
    vs, vv, ds, dv, U0 = Cm.oracle_grids(m1, m2, [100.0])
    p = Cm.oracle_params(m1, m2, N, "DIV")
    U, _, _ = O.solve(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    Ux, _ = O.solve_xp(p, vs[0], vv[0], ds[0], dv[0], U0[0], U0[0])
    assert np.abs(U - Ux).max() < 1e-12 * np.abs(Ux).max()
