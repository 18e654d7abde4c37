#!/usr/bin/env python3
"""Diagnostic (needs a GPU): where one LM iteration of config 4 spends its wall time on the host side.
    python tools/c4_host_profile.py [reps]
Each leg of bench.py's `lm_iteration_case` step timed with a device synchronisation behind it (so the legs do not overlap as they
do in the bench itself: the sum is an upper bound of the iteration), next to the library's own kernel time of the two sweeps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, THETA = 100.0, 0.04, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, 0.8
dev = torch.device("cuda:0")
solver = H.HestonADI(0)
mats = [1.0 + i * 0.25 if i < 8 else 3.0 + (i - 8) * 0.5 for i in range(10)]   # bench.py's surface_points
pts = H.make_calibration_points([S_0 * 0.75 + 1.0 * i for i in range(50)], mats)
cm1, cm2 = 50, 25
cm = (cm1 + 1) * (cm2 + 1)
ks = [p.strike for p in pts]
gh = H.GridViewsBatch.for_strikes(cm1, cm2, S_0, V_0, ks)
grids, U0 = gh.to(dev), torch.from_numpy(gh.call_payoff(ks)).to(dev)
market = torch.tensor([H.market.call_price(S_0, p.strike, R_D, 0.2, p.maturity) for p in pts], dtype=torch.float64, device=dev)
ws = H.DOWorkspace(len(pts), cm, device=dev)
legs = {}
def leg(name, fn):
    torch.cuda.synchronize(); solver.wait_stream()
    t0 = time.perf_counter()
    out = fn()
    solver.wait_stream(); torch.cuda.synchronize()
    legs.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return out
for rep in range(reps + 2):
    if rep == 2: legs.clear()
    J, base = leg("jacobian call", lambda: solver.compute_jacobian_multi_maturity(S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, cm1, cm2, cm, THETA, pts, len(pts), grids, U0))
    legs.setdefault("  its sweep (library timer)", []).append(solver.timing()["sweep_ms"])
    part = leg("partials (device) -> host", lambda: H.lm_partials_device(solver, J, base, market))
    delta = leg("5x5 solve + clamp (host)", lambda: H.lm_solve(part, 0.01))
    new = H.clamp_parameters(KAPPA + delta[0], ETA + delta[1], SIGMA + delta[2], RHO + delta[3], V_0 + delta[4])
    leg("reset of the trial workspace", lambda: ws.U.copy_(U0))
    trial = leg("trial prices call", lambda: solver.compute_base_prices_multi_maturity(S_0, new[4], R_D, R_F, new[3], new[2], new[0], new[1], cm1, cm2, cm, THETA, pts, len(pts), grids, ws))
    legs.setdefault("  its sweep (library timer)", []).append(solver.timing()["sweep_ms"])
    leg("trial error (torch) -> host", lambda: float(((market - trial) ** 2).sum().item()))
tot = 0.0
for k, v in legs.items():
    m = float(np.median(v))
    if not k.startswith("  "): tot += m
    print("%-34s %.3f ms" % (k, m))
print("%-34s %.3f ms (legs synchronised one by one)" % ("sum", tot))
