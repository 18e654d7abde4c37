#!/usr/bin/env python3
"""Instance-resident launch (hadi_team_kernel) against the two-launches-per-step path on small batches of large grids:
python tools/team_ab.py [n_instances ...]   -- wall time per solve, sweep time from the library's events, field difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
if os.environ.get("HADI_LIB"): nat.LIB_PATH = os.path.abspath(os.environ["HADI_LIB"])  # (an alternative build, tools/build_variant.py)
import pde_based_heston_solver_gpu_accelerated_amd as H
dev = torch.device("cuda:0"); s = H.HestonADI(0)
for (m1, m2, N) in ((512, 256, 1000), (256, 128, 500)):
    for n in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 6, 8]:
        ks = [100.0] if n == 1 else [85.0 + 30.0 * k / (n - 1) for k in range(n)]
        g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks); U0 = torch.from_numpy(g.call_payoff(ks)).to(dev); gd = g.to(dev)
        res = {}
        for mode in (0, 1):
            s.set_tuning("team_launch", mode)
            U = torch.empty_like(U0); best, sw = 1e9, 0.0
            for _ in range(4):
                U.copy_(U0); torch.cuda.synchronize(); t = time.perf_counter()
                s.DO_timestepping(m1, m2, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
                dt = time.perf_counter() - t
                if dt < best: best, sw = dt, s.timing()["sweep_ms"]
            res[mode] = (best * 1e3, sw, U.clone(), s.describe_last_sweep()[:48], s.get_tuning("team_launch"))
        d = (res[0][2] - res[1][2]).abs().max().item() / res[0][2].abs().max().item()
        print("%dx%dx%d n=%d  streaming: wall %.2f ms sweep %.2f | team: wall %.2f ms sweep %.2f (%.2f us/step) tuning=%d | rel diff %.1e | %s" % (
            m1, m2, N, n, res[0][0], res[0][1], res[1][0], res[1][1], res[1][1] * 1e3 / N, res[1][4], d, res[1][3]), flush=True)
s.set_tuning("team_launch", -1)
