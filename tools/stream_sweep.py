#!/usr/bin/env python3
"""One stream, two streams, and the automatic choice (hadi_plan_row_idle) over batch sizes and shapes, one process, one box:
    python tools/stream_sweep.py [quick]
Per case: sweep-only point-steps/s (best of 3 after a warm-up) in the three modes, the automatic mode's path, and whether the
automatic choice took the faster of the two forced modes (within 1.5 %)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pde_based_heston_solver_gpu_accelerated_amd as H

S_0, V_0, T, R_D, R_F = 100.0, 0.04, 1.0, 0.025, 0.0
RHO, SIGMA, KAPPA, ETA, THETA = -0.9, 0.3, 1.5, 0.04, 0.8
DIVS = ([0.2, 0.4, 0.6, 0.8], [0.5, 0.3, 0.2, 0.1], [0.02] * 4)
dev = torch.device("cuda:0")
s = H.HestonADI(0)
quick = "quick" in sys.argv[1:]
ring = "ring" in sys.argv[1:]   # only cases whose row pass is the shared ring (the automatic choice stays on one stream there)
CASES = [("c2", 512, 256, 200, n, "EU") for n in ((16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512) if not quick else (64, 160, 192, 256))]
if not quick:
    CASES += [("c3", 256, 128, 200, n, "AM_DIV") for n in (128, 256, 300, 384, 512, 700, 1024)]
    CASES += [("g256", 256, 128, 200, n, "EU") for n in (100, 300, 512, 1024)]
    CASES += [("g200", 200, 100, 200, n, "EU") for n in (200, 700)]
    CASES += [("c5", 1024, 512, 100, n, "EU") for n in (16, 32, 48, 64, 96, 128)]
    CASES += [("am512", 512, 256, 200, n, "AM") for n in (96, 160, 192, 256)]
if "big" in sys.argv[1:]:   # batches of several sub-batches (whole rounds of one instance per CU plus a remainder)
    CASES = [("c2", 512, 256, 100, n, "EU") for n in (320, 384, 448, 512, 600, 640, 768, 1024)] + \
            [("am512", 512, 256, 100, n, "AM") for n in (320, 384, 512, 640)] + [("c5", 1024, 512, 40, n, "EU") for n in (320, 384)]
if ring:
    CASES = [("c2", 512, 256, 200, n, "EU") for n in (16, 24, 32, 48, 96)] + [("c3", 256, 128, 200, n, "AM_DIV") for n in (128, 300, 384)] + \
            [("g256", 256, 128, 200, n, "EU") for n in (100, 200, 300)] + [("g200", 200, 100, 200, n, "EU") for n in (200, 400, 700)] + \
            [("c5", 1024, 512, 100, n, "EU") for n in (16, 24, 48)] + [("am512", 512, 256, 200, 96, "AM")]
for tag, m1, m2, N, n, var in CASES:
    ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
    gh = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks)
    american = var in ("AM", "AM_DIV")
    u0 = torch.from_numpy(gh.put_payoff(ks) if american else gh.call_payoff(ks)).to(dev)
    gd, u = gh.to(dev), torch.empty_like(u0)
    kw = {}
    if american:
        kw = dict(variant=getattr(H, var), U_0=u0, option_type=H.PUT, strikes=ks)
        if var == "AM_DIV":
            kw["dividends"] = H.Dividends(*DIVS)
    res = {}
    for mode in (1, 2, 0):
        s.set_tuning("streams", mode)
        best = 1e30
        for rep in range(4):
            u.copy_(u0)
            s.DO_timestepping(m1, m2, N, T / 1000, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, gd, u, **kw)
            if rep:
                best = min(best, s.timing()["sweep_ms"])
        res[mode] = (n * (m1 + 1) * (m2 + 1) * N / (best * 1e-3), s.describe_last_sweep())
    s.set_tuning("streams", 0)
    took2 = "two streams" in res[0][1]
    better = max(res[1][0], res[2][0])
    verdict = "ok" if res[0][0] >= 0.985 * better else "WORSE than streams=%d by %.1f %%" % (1 if res[1][0] > res[2][0] else 2, 100 * (better / res[0][0] - 1))
    print("%-6s %4dx%-4d n=%-5d one %.4e  two %.4e (%+5.1f %%)  auto %.4e [%s] %s | %s" % (
        tag, m1, m2, n, res[1][0], res[2][0], 100 * (res[2][0] / res[1][0] - 1), res[0][0], "two" if took2 else "one", verdict, res[0][1][:80]), flush=True)
    del gd, u, u0
