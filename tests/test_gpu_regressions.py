"""GPU regression tests of the cases a fuzz campaign flagged (tools/fuzz_parity.py; round 2: gpurun_out/fuzz_a.log seed 2024
cases 378 and 485, gpurun_out/fuzz_small.log seed 5 case 279), adjudicated by the extended-precision build of the oracle
(oracle/heston_oracle_xp.c), and of the device error word behind the bounded pair rendezvous."""
import importlib.util
import os

import numpy as np
import pytest

import pde_based_heston_solver_gpu_accelerated_amd as H

import common as Cm
import fuzz_cases as F

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(Cm.HERE), "tools", "fuzz_parity.py"))
FP = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(FP)


def _worst_instances(r, count):
    """The most ill-conditioned instances of a batch and the instance libhadi and the oracle differ most on."""
    per = np.abs(r["U"] - r["Uo"]).max(axis=1)
    if r["lo"] is not None:
        per = per + np.abs(r["lam"] - r["lo"]).max(axis=1)
    ks = list(np.argsort(-r["ratio"])[:count]) + [int(np.argmax(per))]
    return sorted(set(int(k) for k in ks))


def test_fuzz_seed5_case279_small_american_dividend(solver):
    """'BAD' in round 2 (field 1.16e-7 / lambda_bar 1.24e-6 against the oracle on one instance of 130): instance 101 has S_0
    inserted 7.4e-6 beside a node of a 16-interval s-grid.  The adjudicator showed the defect was libhadi's -- the
    one-node-per-lane cyclic reduction (1.7e-7 from the exact result against the oracle's 4.5e-10) -- fixed by the
    excess-carrying update in hadi_row_step (see tests/test_emu_kernel_logic.py for the CPU twin of this test).  Same batch,
    same kernel choice as the campaign; every ill-conditioned instance within 3x of the oracle's own distance from exact."""
    c = F.case(5, 279, small=True)
    assert F.summary(c) == "AM_DIV m1=16 m2=11 N=9 n=130 r_f=0.03"
    r = FP.run_case(solver, c)
    assert "hadi_small_kernel<1,8,AM>" in r["path"]
    for k in _worst_instances(r, 3):
        a = FP.adjudicate(c, r, k)
        assert a["hadi_U"] < max(3 * a["oracle_U"], 1e-11), (k, a)
        assert a["hadi_lam"] < max(10 * a["oracle_lam"], 1e-9), (k, a)
    ok, j = FP.judge(c, r)
    assert ok, j


def test_fuzz_seed2024_case378_wide_american_put(solver):
    """'BAD' in round 2 by the tool's lambda_bar bound alone (1.23e-7 against the oracle; field 3.4e-10): 977 x 29, two
    wavefronts per row, explicit (U, lambda_bar) pair, instance 96 with neighbouring s-intervals 8021x apart (1.9e-5 beside
    0.155).  Adjudicated: BOTH fp64 solvers sit at cond * eps from the exact result there -- field: libhadi 4.4e-10, oracle
    1.9e-10; lambda_bar (= the unprojected field's error / dt): 1.2e-7 against 1.2e-8.  Scanning the position of such an
    interval over the lanes (same parameters, emulator) the ratio of the two distances ranges over 0.2 .. 5.6: realisations
    of the same round-off, no systematic loss -- unlike the one-node-per-lane defect of case 279 (380x).  Bound: 30x."""
    c = F.case(2024, 378)
    assert F.summary(c) == "AM put m1=977 m2=29 N=12 n=130 r_f=0.00"
    r = FP.run_case(solver, c)
    assert "hadi_pass_a<8,2,4,1,1,AM>" in r["path"]
    for k in _worst_instances(r, 2):
        a = FP.adjudicate(c, r, k)
        assert a["hadi_U"] < max(30 * a["oracle_U"], 1e-11), (k, a)
        assert a["hadi_lam"] < max(30 * a["oracle_lam"], 1e-9), (k, a)
    ok, j = FP.judge(c, r)
    assert ok, j


def test_fuzz_seed2024_case485_fp32_state_dividend_put(solver):
    """'BAD' in round 2 by the fp32-state bound (3.4e-6 against 2e-7 N = 2.4e-6): paired strips, fp32 state, put with
    dividends, 691 x 92.  The checker of this mode is the oracle with the same two roundings per step; the adjudicator (exact
    arithmetic, state rounded where the kernels round it) agrees with that oracle to the last bit here, so the 3.5e-6 are
    libhadi's: float roundings flipped by last-bit fp64 differences (6e-8 each) and grown by the same mechanism that grows
    the mode's own rounding noise -- on this batch the fp32-state oracle is 2.2e-5 from the fp64-state one.  The yardstick is
    therefore that noise, per instance: libhadi's distance from the checker stays below a quarter of the checker's own
    distance from the fp64 result on well-conditioned instances (most differ by 0 or 1 float ulp; the flagged one 3.4e-6
    against a noise of 2.2e-5), and below 4x on the one ill-conditioned instance (s-intervals 30x apart: 2.8e-5 against
    1.1e-5 -- one flip, amplified like the noise itself)."""
    from oracle import oracle as O
    c = F.case(2024, 485)
    assert F.summary(c) == "DIV put f32 m1=691 m2=92 N=12 n=70 r_f=0.00"
    r = FP.run_case(solver, c)
    assert "hadi_pass_a_strip<8,EU,float,2>" in r["path"]
    g, N = r["grids"], c["N"]
    p64 = O.make_params(c["m1"], c["m2"], N, Cm.T / N, Cm.THETA, Cm.R_D, c["r_f"], *c["model"], c["variant"], Cm.DIVS,
                        option_type=O.PUT, strikes=np.array(c["strikes"]))
    U64, _, _ = O.solve_batch(p64, g.Vec_s, g.Vec_v, g.Delta_s, g.Delta_v, r["U0"], r["U0"])
    scale = np.abs(U64).max()
    mine = np.abs(r["U"] - r["Uo"]).max(axis=1) / scale     # libhadi vs the fp32-state checker
    noise = np.abs(r["Uo"] - U64).max(axis=1) / scale       # the fp32 state's own rounding noise, per instance
    ill = r["ratio"] > 30
    assert (mine[~ill] < 0.25 * noise[~ill] + 2e-7 * N).all(), (mine[~ill].max(), noise[~ill].max())
    assert (mine[ill] < 4 * noise[ill] + 2e-7 * N).all(), (mine[ill], noise[ill])
    k = int(np.argmax(np.where(ill, 0.0, mine)))
    a = FP.adjudicate(c, r, k)
    assert a["oracle_U"] < 1e-7 and a["hadi_U"] < 0.25 * noise[k] + 2e-7 * N, (k, a, noise[k])


@pytest.mark.parametrize("strip", [1, 0])
def test_rendezvous_timeout_fails_the_call(solver, strip):
    """The pair rendezvous of the two-wavefront rows (paired strips / shared ring, m1 > 512) polls a bounded number of
    times; running out of polls used to fall through to a solve with stale exchange values and HADI_OK.  Now the kernel
    records it in the handle's device error word and the call fails with HADI_ERR_INTERNAL.  Forced here by a test hook
    that withholds one token per row 1; the next call (hook off) is clean again and bit-identical to a fresh solve."""
    m1, m2, N, strikes = 700, 40, 2, [100.0, 93.0]
    grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, strikes)
    U0 = grids.call_payoff(strikes)

    def run():
        U = U0.copy()
        solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.0, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U)
        return U

    solver.set_tuning("strip", strip)
    try:
        good = run()
        assert ("paired strips" in solver.describe_last_sweep()) == bool(strip)
        solver.set_tuning("debug_fault", 1)
        with pytest.raises(H.HadiError) as ei:
            run()
        assert ei.value.status == 7 and "rendezvous" in str(ei.value)  # HADI_ERR_INTERNAL
        solver.set_tuning("debug_fault", 0)
        assert np.array_equal(run(), good)
    finally:
        solver.set_tuning("debug_fault", 0)
        solver.set_tuning("strip", -1)
