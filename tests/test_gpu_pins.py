"""GPU: libhadi against (i) the data the reference itself holds (convergence sweep and target of src/solver.cpp:1653-1692,
the per-operator acceptance drivers), (ii) the oracle's per-phase intermediates of time step 1 (SURVEY.md 8(c) fixtures
list: A0U, A1U, A2U, Y0, Y1, Y1rhs, U), (iii) the oracle at the BENCHMARKED batch geometries (the plans bench.py times are
the plans checked here, not small-batch relatives), and the extensions that have no reference counterpart (put boundary
data, per-instance V_0, m2 > m1) against their oracle restatements and model-free properties.  Everything goes through
the C ABI."""
import numpy as np
import pytest

import pde_based_heston_solver_gpu_accelerated_amd as H
from oracle import oracle as O

import common as Cm
import test_reference_pins as RP

pytestmark = pytest.mark.gpu

FIELD_RTOL = 1e-10


def _field_err(U, Uo):
    return np.abs(U - Uo).max() / np.abs(Uo).max()


def _batch(m1, m2, strikes, put=False):
    grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, strikes)
    return grids, (grids.put_payoff(strikes) if put else grids.call_payoff(strikes))


# ---- (i) reference-held data ---------------------------------------------------------------------------------------
def test_reference_convergence_sweep_on_gpu(solver):
    """test_convergence (src/solver.cpp:1653-1692) through compute_base_prices: same monotone fall of the relative error
    against the reference's target 8.8948693600540167 as on the oracle, every price equal to the oracle's to 1e-9."""
    errs = []
    for m2 in RP.CONVERGENCE_M2:
        m1, N = 2 * m2, 20
        grids, U0 = _batch(m1, m2, [100.0])
        ws = H.DOWorkspace(1, (m1 + 1) * (m2 + 1))
        ws.U[...] = U0
        price = solver.compute_base_prices(Cm.S_0, Cm.V_0, Cm.T, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, m1, m2,
                                           (m1 + 1) * (m2 + 1), N, Cm.THETA, Cm.T / N, 1, grids, ws)[0]
        assert abs(price - RP.oracle_price(m1, m2, N)) < 1e-9
        errs.append(abs(price - RP.REF_CONVERGENCE_TARGET) / RP.REF_CONVERGENCE_TARGET)
    assert all(a > b for a, b in zip(errs, errs[1:])) and 8e-3 < errs[0] < 9e-3 and errs[-1] < 2.3e-3, errs


def test_config2_price_against_the_semi_analytic_target(solver):
    """BASELINE config 2 (512x256, 1000 steps): 7.3e-5 relative from the reference's semi-analytic target; Richardson
    extrapolation with the N = 500 run removes the first-order time error of theta = 0.8 and leaves the spatial error of
    the 512x256 grid, 3e-5 relative."""
    m1, m2 = 512, 256
    grids, U0 = _batch(m1, m2, [100.0, 100.0])
    U = U0.copy()
    solver.DO_timestepping(m1, m2, 1, 1.0, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                           per_instance={"N_i": [1000, 500], "delta_t_i": [1e-3, 2e-3]})
    node = 181 + 78 * (m1 + 1)
    p1000, p500 = U[0, node], U[1, node]
    assert abs(p1000 - 8.8942192888223310) < 1e-9
    assert abs(p1000 - RP.REF_CONVERGENCE_TARGET) < 8e-5 * RP.REF_CONVERGENCE_TARGET
    assert abs(2 * p1000 - p500 - RP.REF_CONVERGENCE_TARGET) < 4e-5 * RP.REF_CONVERGENCE_TARGET


def test_reference_a2_acceptance_driver_on_gpu(solver):
    """test_device_a2_shuffled_multiple_instances (src/hes_a2_shuffled_kernels.cpp:13-157): 100 instances of the 100x75
    grid, b == 2, x = (I - theta dt A2)^{-1} b through the product's column pass; printed residual
    ||x - theta dt A2 x - b||_2 with A2 x from the oracle's multiply_parallel_s."""
    m1, m2, n = 100, 75, 100
    p, vs, vv, ds, dv = RP.reference_operator_setup(m1, m2)
    grids = H.GridViewsBatch([H.Grid(m1, 800.0, 100.0, 100.0, 20.0, m2, 5.0, 0.04, 0.01)] * n)
    m = (m1 + 1) * (m2 + 1)
    b = 2.0 * np.ones((n, m))
    x = solver.debug_col_solve(m1, m2, 1, p.delta_t, p.theta, p.r_d, 0.0, Cm.RHO, p.sigma, p.kappa, p.eta, grids, b)
    assert "hadi_pass_b" in solver.describe_last_sweep()
    for k in (0, 1, 2, 3, 4, n - 1):  # the reference prints the first five
        Ax, _ = O.operator(p, 2, vs, vv, ds, dv, x[k])
        res = x[k] - p.theta * p.delta_t * Ax - b[k]
        assert np.sqrt((res * res).sum()) < 1e-11
    _, xo = O.operator(p, 2, vs, vv, ds, dv, np.ones(m), b[0])
    assert np.abs(x - xo[None, :]).max() < 1e-12


# ---- (ii) per-phase intermediates of step 1 -------------------------------------------------------------------------
PHASE_SHAPES = [
    # m1, m2, n, strip            kernel family of the row pass / column pass
    (50, 25, 3, -1),            # 1 node per lane, single chunk
    (128, 64, 2, -1),           # 2 nodes per lane, 2 chunks
    (200, 100, 2, -1),          # 4 nodes per lane, 4 chunks
    (512, 256, 2, -1),          # 8 nodes per lane shared ring, 8 chunks
    (512, 256, 2, 1),           # 8 nodes per lane strips
    (256, 128, 2, 1),           # 4 nodes per lane strips
    (700, 300, 1, -1),          # two wavefronts per row, 10 chunks (single-buffer column pass)
]


@pytest.mark.parametrize("m1,m2,n,strip", PHASE_SHAPES)
def test_step1_phases_vs_oracle(solver, m1, m2, n, strip):
    """Y1rhs after the row pass and U after the column pass of time step 1 against the oracle's dump; with them the
    reference's residual acceptance checks on the HIP line solves (hes_a1_kernels.cpp:262-276,
    hes_a2_shuffled_kernels.cpp:142-156): ||x - theta dt A x - b||_2 with A x by the oracle's mat-vec."""
    N, r_f = 20, 0.0
    dt, th = Cm.T / N, Cm.THETA
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    solver.set_tuning("strip", strip)
    try:
        Y = solver.debug_row_pass(m1, m2, N, dt, th, Cm.R_D, r_f, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U0, step=1)
        d = solver.describe_last_sweep()
        U1 = U0.copy()
        solver.set_tuning("small_grid", 0)
        solver.DO_timestepping(m1, m2, 1, dt, th, Cm.R_D, r_f, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U1)
    finally:
        solver.set_tuning("strip", -1)
        solver.set_tuning("small_grid", 1)
    assert ("strip" in d) == (strip == 1)
    p = Cm.oracle_params(m1, m2, N, "EU", r_f=r_f)
    for k in range(n):
        g = [a[k] for a in (grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v)]
        _, _, dump = O.solve(p, *g, U0[k], dump_step=1)
        assert _field_err(Y[k], dump["Y1rhs"]) < FIELD_RTOL
        assert _field_err(U1[k], dump["Unext"]) < FIELD_RTOL
        # A1 line solve of the row pass: Y1 = Y1rhs + theta dt A2U (r_f = 0), residual against Y0rhs
        Y1 = Y[k] + th * dt * dump["A2U"]
        assert _field_err(Y1, dump["Y1"]) < FIELD_RTOL
        A1Y1, _ = O.operator(p, 1, *g, Y1)
        r1 = Y1 - th * dt * A1Y1 - dump["Y0rhs"]
        assert np.sqrt((r1 * r1).sum()) < 1e-9 * np.sqrt((dump["Y0rhs"] ** 2).sum())
        # A2 line solve of the column pass: U1 against the right-hand side the row pass produced
        A2U1, _ = O.operator(p, 2, *g, U1[k])
        r2 = U1[k] - th * dt * A2U1 - Y[k]
        assert np.sqrt((r2 * r2).sum()) < 1e-9 * np.sqrt((Y[k] ** 2).sum())


@pytest.mark.parametrize("m1,m2", [(50, 25), (200, 100), (512, 256), (700, 300)])
def test_explicit_operators_A0_A1_A2_vs_oracle(solver, m1, m2):
    """The fused row pass never materialises A0U, A1U, A2U; one explicit step (theta = 0, dt = 1) returns
    Y = U + A0U + A1U + A2U + b, and parameter choices isolate the three products: rho = 0 removes A0;
    kappa = sigma = 0 leaves A2 = -1/2 r_d I.  Each against the oracle's own dump of that product."""
    K = 100.0
    grids, U0 = _batch(m1, m2, [K])
    rng = np.random.default_rng(5)
    U = U0 + rng.standard_normal(U0.shape)  # (a payoff alone has zero curvature almost everywhere)
    g = [a[0] for a in (grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v)]

    def Y(rho, sigma, kappa):
        return solver.debug_row_pass(m1, m2, 1, 1.0, 0.0, Cm.R_D, 0.0, rho, sigma, kappa, Cm.ETA, grids, U, step=1)[0]

    p = O.make_params(m1, m2, 1, 1.0, 0.0, Cm.R_D, 0.0, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, O.EU)
    _, _, d = O.solve(p, *g, U[0], dump_step=1)
    full, no_a0, only_a1 = Y(Cm.RHO, Cm.SIGMA, Cm.KAPPA), Y(0.0, Cm.SIGMA, Cm.KAPPA), Y(0.0, 0.0, 0.0)
    scale = max(np.abs(d["A1U"]).max(), np.abs(d["A2U"]).max(), np.abs(d["A0U"]).max())
    A0U = full - no_a0
    # with kappa = sigma = 0 the rows 0..m2-2 of A2 are -1/2 r_d I; rows m2-1 and m2 are empty in the reference
    # (hes_a2_shuffled_kernels.hpp:122)
    half = 0.5 * Cm.R_D * U[0].reshape(m2 + 1, m1 + 1).copy()
    half[m2 - 1:] = 0.0
    A1U = only_a1 - U[0] - d["b"] + half.ravel()
    A2U = no_a0 - U[0] - A1U - d["b"]
    for got, want in ((A0U, d["A0U"]), (A1U, d["A1U"]), (A2U, d["A2U"])):
        assert np.abs(got - want).max() < 1e-11 * scale
    assert np.abs(d["A0U"]).max() > 1e-6 * scale and np.abs(d["A2U"]).max() > 1e-6 * scale  # all three are exercised


# ---- (iii) the benchmarked plans ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [256, 64, 512, 160])
def test_headline_batch_geometry_field_vs_oracle(solver, n):
    """bench.py's workload C2 (512x256 grid) at the batch sizes it reports (256 = the headline, 64 / 512 = the batch
    sweep of SURVEY.md 8(d), 160 = a launch that is not a multiple of the CU count): the automatically selected plan,
    full field of every instance against the oracle, a few steps."""
    m1, m2, N = 512, 256, 3 if n <= 256 else 2
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    U = U0.copy()
    solver.DO_timestepping(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U)
    d = solver.describe_last_sweep()
    if n in (256, 512):
        assert "hadi_pass_a_strip<8,EU>" in d
    assert ("2 sub-batches of 256 instances" in d) == (n == 512)
    p = O.make_params(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, O.EU)
    Uo, _, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0)
    errs = np.abs(U - Uo).max(axis=1) / np.abs(Uo).max(axis=1)
    assert errs.max() < FIELD_RTOL, (errs.argmax(), errs.max())


@pytest.mark.parametrize("n,label", [(384, "2 sub-batches of 256 128 instances"), (600, "3 sub-batches of 256 256 88 instances"),
                                     (300, "")])
def test_unequal_sub_batches_vs_oracle(solver, n, label):
    """Whole rounds of one instance per CU plus the remainder (a remainder below a quarter round rides with the last full
    round: 300 stays one launch), each sub-batch with the launch geometry of its own size: fields of the first, last and
    boundary instances against the oracle, and equal to the single-launch run to round-off (not bit for bit: strips that
    walk downwards add the v-neighbours in the opposite order, and the strip partition differs between the geometries)."""
    m1, m2, N = 512, 256, 2
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    U = U0.copy()
    solver.set_tuning("streams", 1)  # (the cut into rounds by itself; the automatic second stream is the next assertion)
    try:
        solver.DO_timestepping(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U)
        d = solver.describe_last_sweep()
    finally:
        solver.set_tuning("streams", 0)
    assert (label in d) if label else ("sub-batches" not in d), d
    # default: the last sub-batch runs as two halves side by side when its row pass would leave a partial round of CUs idle --
    # 300 instances (900 strip blocks on 4 x 256 slots) -- same bits
    W = U0.copy()
    solver.DO_timestepping(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, W)
    d2 = solver.describe_last_sweep()
    # ... and several sub-batches alternate between the two streams (384 = 256 + 128, 600 = 256 + 256 + 88)
    assert "two streams" in d2, d2
    assert np.abs(U - W).max() <= 1e-12 * np.abs(U).max()
    solver.set_tuning("sub_batch", 0)
    try:
        V = U0.copy()
        solver.DO_timestepping(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, V)
    finally:
        solver.set_tuning("sub_batch", 1)
    assert np.abs(U - V).max() <= 1e-12 * np.abs(V).max()
    rows = np.array([0, 1, 255, 256, 257, n - 2, n - 1])
    p = O.make_params(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, O.EU)
    Uo, _, _ = O.solve_batch(p, grids.Vec_s[rows], grids.Vec_v[rows], grids.Delta_s[rows], grids.Delta_v[rows], U0[rows])
    assert _field_err(U[rows], Uo) < FIELD_RTOL


def test_sub_batches_do_not_change_results(solver):
    """512 instances of 512x256 run as two sub-batches of 256 (each round of one block per CU keeps its memory-side cache
    reuse); instances are independent, so the fields are bit-identical to the whole-batch launch, also for American
    options with dividends (per-sub-batch dividend jumps and representation changes)."""
    m1, m2, N, n = 512, 256, 12, 512
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    divs = ([0.002, 0.006], [0.5, 0.3], [0.02, 0.02])
    out = {}
    for variant in (H.EU, H.AM_DIV):
        for sub in (1, 0):
            solver.set_tuning("sub_batch", sub)
            try:
                U, lam = U0.copy(), np.zeros_like(U0)
                solver.DO_timestepping(m1, m2, N, Cm.T / 1000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                                       variant=variant, U_0=U0, lambda_bar=lam if variant == H.AM_DIV else None,
                                       dividends=H.Dividends(*divs) if variant == H.AM_DIV else None)
                assert ("sub-batches" in solver.describe_last_sweep()) == bool(sub)
            finally:
                solver.set_tuning("sub_batch", 1)
            out[(variant, sub)] = (U, lam)
        assert np.array_equal(out[(variant, 1)][0], out[(variant, 0)][0])
        assert np.array_equal(out[(variant, 1)][1], out[(variant, 0)][1])


@pytest.mark.parametrize("m1,m2,N,n,variant,fp32", [(512, 256, 6, 12, "EU", False), (512, 256, 5, 160, "EU", False),
                                                     (256, 128, 30, 37, "AM_DIV", False), (600, 40, 6, 9, "DIV", True)])
def test_two_stream_sweep_matches_one_stream(solver, m1, m2, N, n, variant, fp32):
    """hadi_set_tuning("streams", 2): the batch is cut in two halves that run their time loops side by side on two streams
    (fork / join by events, also inside a captured graph).  Same kernels on independent instances: equal to the one-stream
    run to round-off (the halves have the launch geometry of their own size), per-sub-batch dividend jumps and
    representation changes included; and the first / last instances of both halves against the oracle."""
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    v = getattr(H, variant)
    american = v in (H.AM, H.AM_DIV)
    dt = Cm.T / 500
    out = {}
    for streams in (1, 2, 2, 0):  # (twice on two streams: the second call replays the cached graph where there is one)
        solver.set_tuning("streams", streams)
        try:
            U, lam = U0.copy(), np.zeros_like(U0)
            solver.DO_timestepping(m1, m2, N, dt, Cm.THETA, Cm.R_D, 0.01, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U, variant=v,
                                   U_0=U0, lambda_bar=lam if american else None,
                                   dividends=H.Dividends(*Cm.DIVS) if v in (H.DIV, H.AM_DIV) else None,
                                   state_precision=H.STATE_FP32 if fp32 else H.STATE_FP64)
            path = solver.describe_last_sweep()
            if streams:
                assert ("two streams" in path) == (streams == 2), path
        finally:
            solver.set_tuning("streams", 0)
        if streams == 0:
            # the automatic choice (the default): two streams exactly where the plan's row pass leaves a partial round of CUs
            # idle -- 160 instances of 512x256 (480 strip blocks on 2 x 256 slots) -- and then the very bits of the forced mode
            if (m1, m2, n) == (512, 256, 160):
                assert "two streams" in path, path
            took = 2 if "two streams" in path else 1
            assert np.array_equal(out[took][0], U) and np.array_equal(out[took][1], lam), path
            continue
        if streams in out:
            assert np.array_equal(out[streams][0], U) and np.array_equal(out[streams][1], lam)
        out[streams] = (U, lam)
    tol = 2e-7 * N if fp32 else 1e-12
    assert np.abs(out[1][0] - out[2][0]).max() <= tol * np.abs(out[1][0]).max()
    assert np.abs(out[1][1] - out[2][1]).max() <= 1e-9 * max(1.0, np.abs(out[1][1]).max())
    rows = np.array(sorted({0, (n + 1) // 2 - 1, (n + 1) // 2, n - 1}))
    p = O.make_params(m1, m2, N, dt, Cm.THETA, Cm.R_D, 0.01, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, getattr(O, variant),
                      Cm.DIVS if v in (H.DIV, H.AM_DIV) else None, state_fp32=1 if fp32 else 0)
    Uo, lo, _ = O.solve_batch(p, grids.Vec_s[rows], grids.Vec_v[rows], grids.Delta_s[rows], grids.Delta_v[rows], U0[rows], U0[rows],
                              want_lambda=american)
    assert _field_err(out[2][0][rows], Uo) < (2e-7 * N if fp32 else FIELD_RTOL)


@pytest.mark.parametrize("put", [False, True], ids=["call", "put"])
def test_config3_batch_geometry_field_vs_oracle(solver, put):
    """BASELINE config 3 as benchmarked: 512 American options with dividends on 256x128, 30 of the 500 steps (step size of
    the full run, so the dividend dates 0.02..0.06 fall inside) -- call payoff with the reference's boundary data, and
    put payoff with HADI_PUT boundary data -- U and lambda_bar of every instance against the oracle."""
    m1, m2, N, n = 256, 128, 30, 512
    dt = Cm.T / 500
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes, put=put)
    divs = ([0.006, 0.02, 0.04, 0.05], [0.5, 0.3, 0.2, 0.1], [0.02] * 4)
    U, lam = U0.copy(), np.zeros_like(U0)
    solver.DO_timestepping(m1, m2, N, dt, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                           variant=H.AM_DIV, U_0=U0, lambda_bar=lam, dividends=H.Dividends(*divs),
                           option_type=H.PUT if put else H.CALL, strikes=strikes if put else None)
    assert "AM-P" in solver.describe_last_sweep()
    p = O.make_params(m1, m2, N, dt, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, O.AM_DIV, divs,
                      option_type=O.PUT if put else O.CALL, strikes=np.array(strikes) if put else None)
    Uo, lo, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0, U0, want_lambda=True)
    errs = np.abs(U - Uo).max(axis=1) / np.abs(Uo).max(axis=1)
    assert errs.max() < FIELD_RTOL, (errs.argmax(), errs.max())
    assert np.abs(lam - lo).max() <= 1e-8 * max(1.0, np.abs(lo).max())
    assert lo.max() > 0


@pytest.mark.parametrize("fp32", [True, False])
def test_config5_batch_geometry_vs_oracle(solver, fp32):
    """BASELINE config 5 as benchmarked: 64 instances of 1024x512, 3 steps -- with the fp32 state against the oracle with
    the same roundings, and with the fp64 state (the one that meets the 1e-6 tolerance; `bench.py --workload c5 --state
    fp64`).  Both run the paired strips (two wavefronts per v-row) and the 16-chunk column pass the bench line times."""
    m1, m2, N, n = 1024, 512, 3, 64
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    U = U0.copy()
    solver.DO_timestepping(m1, m2, N, Cm.T / 2000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                           state_precision=H.STATE_FP32 if fp32 else H.STATE_FP64)
    d = solver.describe_last_sweep()
    assert ("hadi_pass_a_strip<8,EU,float,2>" if fp32 else "hadi_pass_a_strip<8,EU,double,2>") in d and "hadi_pass_b1<16" in d
    p = O.make_params(m1, m2, N, Cm.T / 2000, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, O.EU, None,
                      state_fp32=1 if fp32 else 0)
    Uo, _, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0)
    errs = np.abs(U - Uo).max(axis=1) / np.abs(Uo).max(axis=1)
    assert errs.max() < (2e-7 * N if fp32 else FIELD_RTOL), (errs.argmax(), errs.max())


def test_config4_surface_jacobian_vs_oracle(solver):
    """BASELINE config 4 as benchmarked: the 500-option surface (50 strikes x 10 maturities, N_m = max(20, 20 T_m),
    heston_calibration.cpp:2485-2531) on the 50x25 grid -- one flattened Jacobian sweep of 3000 instances with
    per-instance (N, dt) -- against the oracle's six sequential solves per option."""
    m1, m2 = 50, 25
    mats = [1.0 + i * 0.25 if i < 8 else 3.0 + (i - 8) * 0.5 for i in range(10)]
    pts = H.make_calibration_points([Cm.S_0 * 0.75 + 1.0 * i for i in range(50)], mats)
    assert len(pts) == 500
    grids, U0 = _batch(m1, m2, [p.strike for p in pts])
    J, base = solver.compute_jacobian_multi_maturity(Cm.S_0, Cm.V_0, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA,
                                                     m1, m2, (m1 + 1) * (m2 + 1), Cm.THETA, pts, len(pts), grids, U0)
    Jo, baseo = Cm.OracleSolver().compute_jacobian_multi_maturity(Cm.S_0, Cm.V_0, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA,
                                                                  Cm.KAPPA, Cm.ETA, m1, m2, (m1 + 1) * (m2 + 1), Cm.THETA,
                                                                  pts, len(pts), grids, U0)
    assert np.abs(base - baseo).max() < 1e-9
    assert np.abs(J - Jo).max() < 2e-4, np.abs(J - Jo).max()


# ---- put boundary data (no reference counterpart) ------------------------------------------------------------------------
PUT_CASES = [
    # m1, m2, N, n, variant, tuning
    (50, 25, 20, 3, H.EU, {}),                       # LDS-resident small-grid kernel
    (50, 25, 20, 3, H.AM_DIV, {}),
    (50, 25, 20, 2, H.AM_DIV, {"small_grid": 0}),    # streaming kernels, 1 node per lane
    (128, 64, 10, 2, H.AM, {}),                      # 2 nodes per lane, P representation
    (256, 128, 25, 2, H.AM_DIV, {"american_p": 0}),  # explicit (U, lambda_bar) pair, dividend steps
    (512, 256, 8, 2, H.EU, {}),                      # shared ring at 8 nodes per lane
    (512, 256, 8, 2, H.AM, {"strip": 1}),            # strips, P representation
    (300, 140, 6, 2, H.DIV, {"strip": 1}),           # strips, dividends
    (700, 300, 4, 1, H.AM, {}),                      # two wavefronts per row
]


@pytest.mark.parametrize("m1,m2,N,n,variant,tuning", PUT_CASES)
def test_put_boundary_data_vs_oracle(solver, m1, m2, N, n, variant, tuning):
    """HADI_PUT on every kernel family against the oracle's put restatement (b1 == 0, b2 = -1/2 r_d K with time factor
    e^{-r_d t}, reaction term on the i = 0 row, ex-dividend spot <= 0 taking the s = 0 value), full field and lambda_bar."""
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes, put=True)
    U, lam = U0.copy(), np.zeros_like(U0)
    american = variant in (H.AM, H.AM_DIV)
    div = Cm.DIVS if variant in (H.DIV, H.AM_DIV) else None
    for k, v in tuning.items():
        solver.set_tuning(k, v)
    try:
        solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.01, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                               variant=variant, U_0=U0, lambda_bar=lam if american else None,
                               dividends=H.Dividends(*div) if div else None, option_type=H.PUT, strikes=strikes)
    finally:
        for k in tuning:
            solver.set_tuning(k, {"strip": -1, "small_grid": 1, "american_p": 1}[k])
    p = O.make_params(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.01, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, variant, div,
                      option_type=O.PUT, strikes=np.array(strikes))
    Uo, lo, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0, U0, want_lambda=True)
    assert _field_err(U, Uo) < FIELD_RTOL
    if american:
        assert np.abs(lam - lo).max() <= 1e-8 * max(1.0, np.abs(lo).max())


@pytest.mark.parametrize("m1,m2,N", [(50, 25, 20), (100, 50, 200), (200, 100, 40), (256, 128, 500), (512, 256, 1000)])
def test_put_call_parity_and_early_exercise(solver, m1, m2, N):
    """European put against the European call of the same instance by put-call parity at the price node.  The Douglas
    step discounts with g = its amplification factor of du/dt = -r_d u instead of e^{-r_d dt}; against that discrete
    forward S_0 - K g^N parity holds to 1e-6 on every grid of the ladder (observed <= 1e-8), against the continuous
    forward S_0 e^{-r_f T} - K e^{-r_d T} to the first-order time error 0.3 r_d^2 K T dt.  The American put dominates the
    European one and its payoff."""
    K = 100.0
    grids, C0 = _batch(m1, m2, [K])
    P0 = grids.put_payoff([K])
    args = (m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids)
    Uc, Up, Ua = C0.copy(), P0.copy(), P0.copy()
    solver.DO_timestepping(*args, Uc)
    solver.DO_timestepping(*args, Up, option_type=H.PUT, strikes=[K])
    solver.DO_timestepping(*args, Ua, variant=H.AM, U_0=P0, option_type=H.PUT, strikes=[K])
    g0 = H.Grid(m1, 8 * K, Cm.S_0, K, K / 5, m2, 5.0, Cm.V_0, 5.0 / 500)
    node = g0.find_s_index(Cm.S_0) + g0.find_v0_index(Cm.V_0) * (m1 + 1)
    th, dt, r = Cm.THETA, Cm.T / N, Cm.R_D
    g = ((1 - dt * r + th * dt * 0.5 * r) / (1 + th * dt * 0.5 * r) + th * dt * 0.5 * r) / (1 + th * dt * 0.5 * r)
    c, p, a = Uc[0, node], Up[0, node], Ua[0, node]
    # (s - K g^N is not an exact discrete solution next to the quirky edges -- b1 at m1*(j+1), the two empty top rows of
    # A2 -- and the coarsest grid feels that at the price node: 1.9e-6 on 50x25, <= 1e-7 from 100x50 up)
    assert abs((c - p) - (Cm.S_0 - K * g ** N)) < (3e-6 if m1 == 50 else 1e-6)
    assert abs((c - p) - (Cm.S_0 - K * np.exp(-r * Cm.T))) < 0.35 * r * r * K * Cm.T * dt + 1e-6
    assert a > p and a >= P0[0, node] and (Ua >= P0 - 1e-12).all()
    # against the closed form: the put carries the same discretisation error as the call (they differ by the forward)
    put_exact = RP.REF_CONVERGENCE_TARGET - Cm.S_0 + K * np.exp(-r * Cm.T)
    assert abs((p - put_exact) - (c - RP.REF_CONVERGENCE_TARGET)) < 0.35 * r * r * K * Cm.T * dt + 1e-6


# ---- per-instance V_0, device v-grid rebuild, m2 > m1 ---------------------------------------------------------------------
def test_per_instance_V0_device_rebuild(solver):
    """N2: every instance's v-grid is rebuilt on the device for its OWN V_0 (grid_pod.hpp:25-73; the reference rebuilds per
    team).  Each instance must equal an individual oracle compute_base_prices with that V_0; the Jacobian's v0 column is
    formed from the per-instance V_0 + eps grids."""
    m1, m2, N = 50, 25, 20
    strikes = [90.0, 100.0, 110.0, 95.0, 105.0]
    v0s = [0.04, 0.0225, 0.09, 0.0123, 0.25]
    grids, U0 = _batch(m1, m2, strikes)
    ws = H.DOWorkspace(len(strikes), (m1 + 1) * (m2 + 1))
    ws.U[...] = U0
    args = (Cm.T, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, m1, m2, (m1 + 1) * (m2 + 1), N, Cm.THETA, Cm.T / N,
            len(strikes), grids)
    prices = solver.compute_base_prices(Cm.S_0, -1.0, *args, ws, per_instance={"V_0_i": v0s})
    J, base = solver.compute_jacobian(Cm.S_0, -1.0, *args, U0, per_instance={"V_0_i": v0s})
    p = Cm.oracle_params(m1, m2, N, "EU")
    for k, v0 in enumerate(v0s):
        sl = slice(k, k + 1)
        want, _ = O.base_prices(p, Cm.S_0, v0, grids.Vec_s[sl], grids.Vec_v[sl], grids.Delta_s[sl], grids.Delta_v[sl], U0[sl])
        Jo, bo = O.jacobian(p, Cm.S_0, v0, grids.Vec_s[sl], grids.Vec_v[sl], grids.Delta_s[sl], grids.Delta_v[sl], U0[sl])
        assert abs(prices[k] - want[0]) < 1e-9 and abs(base[k] - bo[0]) < 1e-9
        assert np.abs(J[k] - Jo[0]).max() < 2e-4
    # host-built grids (glibc sinh/asinh, one shared V_0) and device-built grids agree to round-off; V_0_i needs the device
    ws.U[...] = U0
    dev = solver.compute_base_prices(Cm.S_0, Cm.V_0, *args, ws)
    solver.set_tuning("device_vgrid", 0)
    try:
        ws.U[...] = U0
        host = solver.compute_base_prices(Cm.S_0, Cm.V_0, *args, ws)
        with pytest.raises(H.HadiError):
            solver.compute_base_prices(Cm.S_0, Cm.V_0, *args, ws, per_instance={"V_0_i": v0s})
    finally:
        solver.set_tuning("device_vgrid", 1)
    assert np.abs(dev - host).max() < 1e-12


@pytest.mark.parametrize("m1,m2,N,variant,name", [(20, 30, 4, H.AM, "AM"), (20, 50, 4, H.EU, "EU"), (40, 70, 3, H.AM, "AM"), (70, 100, 3, H.EU, "EU"),
                                                  (64, 65, 3, H.AM_DIV, "AM_DIV"), (300, 400, 2, H.EU, "EU")])
def test_more_v_nodes_than_s_nodes(solver, m1, m2, N, variant, name):
    """m2 > m1 (rejected in round 1): the reference's b1 index m1*(j+1) then lands twice on the v-rows k*m1 (columns 0 and
    m1); the row table carries both.  r_f != 0 so that the b1 terms matter."""
    strikes = Cm.strikes_for(2)
    grids, U0 = _batch(m1, m2, strikes)
    U, lam = U0.copy(), np.zeros_like(U0)
    american = variant in (H.AM, H.AM_DIV)
    div = H.Dividends(*Cm.DIVS) if variant == H.AM_DIV else None
    for small in (1, 0):
        U[...] = U0
        solver.set_tuning("small_grid", small)
        try:
            solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.02, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                                   variant=variant, U_0=U0, lambda_bar=lam if american else None, dividends=div)
        finally:
            solver.set_tuning("small_grid", 1)
        p = Cm.oracle_params(m1, m2, N, name, r_f=0.02)
        Uo, lo, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0, U0, want_lambda=True)
        assert _field_err(U, Uo) < FIELD_RTOL


@pytest.mark.parametrize("m1,m2,N,variant,name,tuning", [
    (300, 400, 2, H.EU, "EU", {"strip": 1}),        # strips, 8 nodes per lane; rows 300 carries b1 at columns 0 AND m1
    (300, 400, 2, H.AM, "AM", {"strip": 0}),        # shared ring, same rows
    (150, 160, 3, H.EU, "EU", {"strip": 1}),        # strips, 4 nodes per lane
    (100, 130, 3, H.DIV, "DIV", {"strip": 1}),      # strips, 2 nodes per lane
    (12, 30, 4, H.EU, "EU", {"small_seq": 1}),      # one wavefront per instance, rows 12 and 24 carry both entries
    (12, 30, 4, H.DIV, "DIV", {"small_seq": 0}),    # block kernel
    (30, 30, 4, H.EU, "EU", {"small_seq": 1}),      # m2 == m1: the last v-row r = m1 has its b1 entry at column 0 only
    (64, 64, 3, H.AM, "AM", {}),
    (520, 527, 2, H.EU, "EU", {"strip": 1}),        # paired strips: column 0 in the low half, column m1 in the high half
    (520, 527, 2, H.EU, "EU", {"strip": 0}),
])
def test_two_b1_entries_per_row_on_every_row_kernel(solver, m1, m2, N, variant, name, tuning):
    """The rows k*m1 of a grid with m2 >= m1 (RC_B1COL + HADI_B1_BOTH: b1 at column 0 and at column m1, or at column 0 alone
    when m2 == m1) on each row-pass family by name: strips at 8 / 4 / 2 nodes per lane, paired strips, the shared ring, the
    two LDS-resident kernels.  r_f != 0 so that the b1 terms carry weight."""
    strikes = Cm.strikes_for(3)
    grids, U0 = _batch(m1, m2, strikes)
    U, lam = U0.copy(), np.zeros_like(U0)
    american = variant in (H.AM, H.AM_DIV)
    div = H.Dividends(*Cm.DIVS) if variant in (H.DIV, H.AM_DIV) else None
    for k, v in tuning.items():
        solver.set_tuning(k, v)
    try:
        solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.02, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U,
                               variant=variant, U_0=U0, lambda_bar=lam if american else None, dividends=div)
        path = solver.describe_last_sweep()
    finally:
        for k in tuning:
            solver.set_tuning(k, -1)
    if "strip" in tuning:
        assert ("strip" in path) == bool(tuning["strip"]), path
    if "small_seq" in tuning:
        assert ("hadi_small_seq_kernel" in path) == bool(tuning["small_seq"]), path
    p = Cm.oracle_params(m1, m2, N, name, r_f=0.02)
    Uo, lo, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0, U0, want_lambda=True)
    assert _field_err(U, Uo) < FIELD_RTOL
    if american:
        assert np.abs(lam - lo).max() < 1e-8 * max(1.0, np.abs(lo).max())


def test_reciprocal_of_the_line_solves_is_within_16_ulp(solver):
    """hadi_rcp (v_rcp_f64 + ONE Newton step; round 2 dropped the third-order step to save an FMA on each of the 15 reciprocals
    of a row): its error against the correctly rounded 1/x over the magnitudes the pivots take -- 1 (dt -> 0) to 1e9 (a
    7e-6-wide s-interval) -- both signs, and near powers of two where v_rcp_f64's table switches.  Measured on MI355X: up to
    10 ulp (v_rcp_f64 is good to ~2^-24.5 on some arguments, one Newton step squares that) -- 2e-15 relative, a backward
    error of the same size as the rounding of the pivot's own operands."""
    rng = np.random.default_rng(7)
    x = np.concatenate([10.0 ** rng.uniform(-3, 10, 200000) * rng.choice([-1.0, 1.0], 200000),
                        1.0 + rng.uniform(0, 1e-6, 20000), 2.0 ** rng.integers(-20, 30, 20000) * (1.0 + rng.uniform(-1e-9, 1e-9, 20000))])
    got = solver.debug_rcp(x)
    exact = 1.0 / x
    ulp = np.spacing(np.abs(exact))
    assert (np.abs(got - exact) / ulp).max() <= 16.0


# ---- input ordering, device-side LM reduction --------------------------------------------------------------------------------
def test_device_inputs_are_ordered_after_the_torch_stream(solver):
    """The handle runs on its own non-blocking stream: tensors written by torch ops enqueued right before the call (no
    synchronize) must be complete when the library reads them (hadi_wait_stream in the host mirror).  A large fill + copy
    chain on torch's stream, then the solve; bit-identical to the host-memory path."""
    import torch
    m1, m2, N, n = 512, 256, 2, 96
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    U_host = U0.copy()
    solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U_host)
    dev = torch.device("cuda:0")
    gd = grids.to(dev)
    U0d = torch.from_numpy(U0).to(dev)
    U = torch.empty_like(U0d)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device=dev)  # 512 MB: keeps the stream busy for ~0.2 ms
    torch.cuda.synchronize()
    for _ in range(3):
        U.fill_(float("nan"))
        junk.fill_(1.0)
        junk.mul_(2.0)
        U.copy_(U0d)  # still in flight when the launcher is entered
        solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, gd, U)
        assert np.array_equal(U.cpu().numpy(), U_host)


def test_lm_partials_on_the_device(solver):
    """J^T J, J^T r, sum r^2 reduced on the GPU from device-resident Jacobian rows (what hadi_compute_jacobian leaves in
    HBM): equal to the host loop to round-off, additive over shards, empty shard -> zeros."""
    import torch
    dev = torch.device("cuda:0")
    m1, m2, N = 50, 25, 20
    strikes = [85.0 + 0.5 * k for k in range(61)]
    grids, U0 = _batch(m1, m2, strikes)
    gd, U0d = grids.to(dev), torch.from_numpy(U0).to(dev)
    J, base = solver.compute_jacobian(Cm.S_0, Cm.V_0, Cm.T, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, m1, m2,
                                      (m1 + 1) * (m2 + 1), N, Cm.THETA, Cm.T / N, len(strikes), gd, U0d)
    assert J.is_cuda and base.is_cuda
    market = base + 0.05 * torch.sin(torch.arange(len(strikes), device=dev, dtype=torch.float64))
    got = H.lm_partials_device(solver, J, base, market)
    want = H.lm_partials(J.cpu().numpy(), (market - base).cpu().numpy())
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    a = H.lm_partials_device(solver, J[:20].contiguous(), base[:20].contiguous(), market[:20].contiguous())
    b = H.lm_partials_device(solver, J[20:].contiguous(), base[20:].contiguous(), market[20:].contiguous())
    assert np.allclose(a + b, got, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    z = H.lm_partials_device(solver, J[:0].contiguous(), base[:0].contiguous(), market[:0].contiguous())
    assert np.all(z == 0.0)


def test_plan_cost_model_constants_are_tuning_keys(solver):
    """The strips-or-ring decision at 8 nodes per lane compares two modelled times whose constants were measured on one box
    (hadi_plan.h); they are adjustable per handle.  Making a strip row step 10x dearer sends a 256-instance batch to the
    shared ring, the default brings the strips back; results agree to round-off either way."""
    m1, m2, N, n = 512, 256, 2, 256
    strikes = Cm.strikes_for(n)
    grids, U0 = _batch(m1, m2, strikes)
    out = {}
    default = solver.get_tuning("model_strip_row_ns")
    assert default == 2800
    for val in (default * 10, default):
        solver.set_tuning("model_strip_row_ns", val)
        U = U0.copy()
        solver.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, Cm.R_F, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids, U)
        out[val] = (U, solver.describe_last_sweep())
    assert "hadi_pass_a<8,1" in out[default * 10][1] and "hadi_pass_a_strip<8,EU>" in out[default][1]
    # (two row kernels = two operation orders: measured 1.4e-12 of max|U| after N steps)
    assert np.abs(out[default][0] - out[default * 10][0]).max() < 1e-11 * np.abs(out[default][0]).max()
    with pytest.raises(H.HadiError):
        solver.set_tuning("model_no_such_constant", 5)
