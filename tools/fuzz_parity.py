#!/usr/bin/env python3
"""Diagnostic (needs a GPU): random grid shapes / batch sizes / variants / parameters through libhadi against the oracle.

    python tools/fuzz_parity.py SEED COUNT          (FUZZ_SMALL=1: LDS-resident shapes; FUZZ_WIDE=1: two wavefronts per row;
                                                     FUZZ_ONLY=i,j: only these case indices; FUZZ_PAIRS=1: grids with
                                                     128 < m1 <= 256 forced onto hadi_pass_a_pairs)

The cases come from tests/fuzz_cases.py (case INDEX of seed SEED is always the same problem; flagged cases become
regression tests in tests/test_gpu_regressions.py).  Judged per instance:
  well-conditioned instances     field 1e-10 of max|U|, lambda_bar 1e-8 of max(1, |lambda_bar|), against the oracle
  ill-conditioned instances      (neighbouring s-intervals differing by > 30x: S_0 inserted right beside a node; 1/ds^2
                                 coefficients of 1e6+ amplify the round-off of ANY fp64 solver -- at a ratio of 6e6 the ORACLE
                                 is 6.6e-7 from the exact result) by the extended-precision adjudicator (oracle.solve_xp), EVERY
                                 such instance whose difference from the oracle leaves the well-conditioned band: libhadi
                                 may be at most 30x further from the exact result than the fp64 oracle is (plus a sanity
                                 cap of 1e-4 against the oracle)
  fp32 state                     against the oracle with the same roundings, with that instance's own fp32 noise as the
                                 yardstick (last-bit fp64 differences flip float roundings, and the flips grow like the
                                 noise): |libhadi - oracle32| <= 1.5 |oracle32 - oracle64| + 2e-7 N per well-conditioned
                                 instance, 10x on ill-conditioned ones
Every BAD line is followed by the adjudicator's verdict on its worst instance.

THE RULES ABOVE ARE FROZEN (round 4).  They were fitted twice to cases that tripped them (the fp32 rule for ill-conditioned
instances went 4x -> 10x in round 3); a bound that moves whenever it trips is not a bound.  From here on a BAD line is a
FINDING -- it becomes a regression test with its adjudicated numbers and, if the adjudicator blames libhadi, a defect to fix --
never a new constant.  Since round 4 every ill-conditioned instance outside the well-conditioned band is adjudicated, not only
the worst one of its case."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
if os.environ.get("HADI_LIB"): nat.LIB_PATH = os.path.abspath(os.environ["HADI_LIB"])
import pde_based_heston_solver_gpu_accelerated_amd as H
from oracle import oracle as O
import common as Cm
import fuzz_cases as F


def run_case(s, c):
    """Solve case c on libhadi and on the oracle.  Returns a dict with both results and the per-instance inputs."""
    m1, m2, N, variant, put, f32, strikes = c["m1"], c["m2"], c["N"], c["variant"], c["put"], c["f32"], c["strikes"]
    grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, strikes)
    U0 = grids.put_payoff(strikes) if put else grids.call_payoff(strikes)
    U, lam = U0.copy(), np.zeros_like(U0)
    div = H.Dividends(*Cm.DIVS) if variant in (H.DIV, H.AM_DIV) else None
    defaults = {"american_p": 1, "strip": -1, "small_seq": -1, "pair_strips": -1}
    if os.environ.get("FUZZ_PAIRS") and 128 < m1 <= 256:  # (not part of the generator's stream: the cases stay the same problems)
        c["tuning"]["strip"], c["tuning"]["pair_strips"] = 1, 1
    for k, v in c["tuning"].items(): s.set_tuning(k, v)
    try:
        s.DO_timestepping(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, c["r_f"], *c["model"], grids, U, variant=variant, U_0=U0,
                          lambda_bar=lam if variant in (H.AM, H.AM_DIV) else None, dividends=div,
                          option_type=H.PUT if put else H.CALL, strikes=strikes if put else None,
                          state_precision=H.STATE_FP32 if f32 else H.STATE_FP64)
        path = s.describe_last_sweep()
    finally:
        for k in c["tuning"]: s.set_tuning(k, defaults[k])
    p = O.make_params(m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, c["r_f"], *c["model"], variant, Cm.DIVS if div is not None else None,
                      option_type=O.PUT if put else O.CALL, strikes=np.array(strikes) if put else None, state_fp32=1 if f32 else 0)
    Uo, lo, _ = O.solve_batch(p, grids.Vec_s, grids.Vec_v, grids.Delta_s, grids.Delta_v, U0, U0, want_lambda=True)
    ds = np.asarray(grids.Delta_s)
    ratio = np.maximum(ds[:, 1:] / ds[:, :-1], ds[:, :-1] / ds[:, 1:]).max(axis=1)
    return dict(U=U, lam=lam, Uo=Uo, lo=lo, U0=U0, grids=grids, params=p, ratio=ratio, path=path)


def adjudicate(c, r, k):
    """Instance k through the extended-precision adjudicator: relative distances of libhadi and of the fp64 oracle from the
    exact result, for the field and for lambda_bar (None for European variants)."""
    g = r["grids"]
    Ux, lx = O.solve_xp(r["params"], g.Vec_s[k], g.Vec_v[k], g.Delta_s[k], g.Delta_v[k], r["U0"][k], r["U0"][k],
                        strike=c["strikes"][k] if c["put"] else None)
    sc = np.abs(Ux).max()
    out = dict(k=k, hadi_U=np.abs(r["U"][k] - Ux).max() / sc, oracle_U=np.abs(r["Uo"][k] - Ux).max() / sc, hadi_lam=None, oracle_lam=None)
    if lx is not None:
        sl = max(1.0, np.abs(lx).max())
        out["hadi_lam"], out["oracle_lam"] = np.abs(r["lam"][k] - lx).max() / sl, np.abs(r["lo"][k] - lx).max() / sl
    return out


def judge(c, r):
    """(ok, numbers) by the rules in the module docstring."""
    N, f32 = c["N"], c["f32"]
    ill = r["ratio"] > 30
    scale = np.abs(r["Uo"]).max()
    per = np.abs(r["U"] - r["Uo"]).max(axis=1) / scale
    err = per[~ill].max() if (~ill).any() else 0.0
    err_ill = per[ill].max() if ill.any() else 0.0
    lerr = lerr_ill = 0.0
    if r["lo"] is not None:
        lper = np.abs(r["lam"] - r["lo"]).max(axis=1) / max(1.0, np.abs(r["lo"]).max())
        lerr = lper[~ill].max() if (~ill).any() else 0.0
        lerr_ill = lper[ill].max() if ill.any() else 0.0
    ok = all(np.isfinite(x) for x in (err, err_ill, lerr, lerr_ill))
    if f32:  # the instance's own fp32-state noise is the yardstick
        p64 = O.make_params(c["m1"], c["m2"], N, Cm.T / N, Cm.THETA, Cm.R_D, c["r_f"], *c["model"], c["variant"],
                            Cm.DIVS if c["variant"] in (H.DIV, H.AM_DIV) else None, option_type=O.PUT if c["put"] else O.CALL,
                            strikes=np.array(c["strikes"]) if c["put"] else None)
        g = r["grids"]
        U64, _, _ = O.solve_batch(p64, g.Vec_s, g.Vec_v, g.Delta_s, g.Delta_v, r["U0"], r["U0"])
        noise = np.abs(r["Uo"] - U64).max(axis=1) / scale
        # (ill-conditioned instances: the flips are amplified like the noise itself, and more erratically -- 4.0x at an interval
        # ratio of 30 (seed 2024 #485), 7.7x at 482 (seed 331 #358): 10x)
        ok = ok and bool((per[~ill] < 1.5 * noise[~ill] + 2e-7 * N).all()) and bool((per[ill] < 10 * noise[ill] + 2e-7 * N).all())
    else:
        ok = ok and err < 1e-10 and lerr < 1e-8 and err_ill < 1e-4 and lerr_ill < 1e-3
    verdict = None
    if ill.any() and not f32:
        # EVERY ill-conditioned instance outside the well-conditioned band (field 1e-10, lambda_bar 1e-8) goes through the
        # adjudicator -- a defect must not hide behind a worse neighbour of its batch; the rest are inside the band anyway.
        # The verdict printed is the worst one's.
        lp = lper if r["lo"] is not None else np.zeros_like(per)
        todo = [int(k) for k in np.where(ill)[0] if per[k] >= 1e-10 or lp[k] >= 1e-8]
        if not todo:
            todo = [int(np.where(ill)[0][np.argmax(per[ill] + lp[ill])])]
        for k in todo:
            v = adjudicate(c, r, k)
            # (both distances are realisations of cond * eps: scanning the position of a 2e-5-wide interval over the lanes their
            # ratio ranges over 0.2 .. 10; the one-node-per-lane defect this check caught in round 3 was 380x)
            ok = ok and v["hadi_U"] < max(30 * v["oracle_U"], 1e-11)
            if v["hadi_lam"] is not None:
                ok = ok and v["hadi_lam"] < max(30 * v["oracle_lam"], 1e-9)
            if verdict is None or v["hadi_U"] / max(v["oracle_U"], 1e-300) > verdict["hadi_U"] / max(verdict["oracle_U"], 1e-300):
                verdict = v
        verdict["adjudicated"] = len(todo)
    return ok, dict(err=err, err_ill=err_ill, lerr=lerr, lerr_ill=lerr_ill, per=per, verdict=verdict)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    small, wide = bool(os.environ.get("FUZZ_SMALL")), bool(os.environ.get("FUZZ_WIDE"))
    s = H.HestonADI(0)
    worst, bad = 0.0, 0
    only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]  # FUZZ_ONLY=i,j: just these case indices
    for c in F.cases(seed, count, small, wide):
        if only and c["index"] not in only: continue
        if os.environ.get("FUZZ_PAIRS") and not (128 < c["m1"] <= 256): continue
        r = run_case(s, c)
        ok, j = judge(c, r)
        if not c["f32"]: worst = max(worst, j["err"])
        v = j["verdict"]
        vs = "" if v is None else " | xp %d inst, worst %d: hadi %.1e oracle %.1e%s" % (
            v.get("adjudicated", 1), v["k"], v["hadi_U"], v["oracle_U"], "" if v["hadi_lam"] is None else " lam %.1e / %.1e" % (v["hadi_lam"], v["oracle_lam"]))
        print("%s %3d %s err=%.2e ill=%.2e lam_err=%.2e lam_ill=%.2e%s | %s" % (
            "ok " if ok else "BAD", c["index"], F.summary(c), j["err"], j["err_ill"], j["lerr"], j["lerr_ill"], vs, r["path"][:70]), flush=True)
        if not ok:
            bad += 1
            k = int(np.argmax(j["per"]))
            a = adjudicate(c, r, k)  # (for an fp32 state the adjudicator rounds the state where the kernels do)
            print("    adjudicator, worst instance %d (ds ratio %.1f): |hadi - exact| %.2e, |oracle - exact| %.2e%s" % (
                k, r["ratio"][k], a["hadi_U"], a["oracle_U"],
                "" if a["hadi_lam"] is None else "; lambda_bar %.2e / %.2e" % (a["hadi_lam"], a["oracle_lam"])), flush=True)
    print("%d bad of %d, worst fp64 field error %.2e" % (bad, count, worst))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
