#!/usr/bin/env python3
"""Where one LM iteration on the 500-option surface (bench.py --workload c4) spends its time: per launcher the library's own
setup / sweep / finish events next to the host wall clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import pde_based_heston_solver_gpu_accelerated_amd as H
S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, THETA = bench.S_0, bench.V_0, bench.R_D, bench.R_F, bench.RHO, bench.SIGMA, bench.KAPPA, bench.ETA, bench.THETA
m1, m2 = 50, 25; m = (m1 + 1) * (m2 + 1)
dev = torch.device("cuda:0"); s = H.HestonADI(0)
pts = bench.surface_points(H); ks = [p.strike for p in pts]
g = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks); U0 = torch.from_numpy(g.call_payoff(ks)).to(dev); gd = g.to(dev)
market = torch.tensor([H.market.call_price(S_0, p.strike, R_D, 0.2, p.maturity) for p in pts], dtype=torch.float64, device=dev)
ws = H.DOWorkspace(len(pts), m, device=dev)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    J, base = s.compute_jacobian_multi_maturity(S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, m1, m2, m, THETA, pts, len(pts), gd, U0)
    t1 = time.perf_counter(); tj = s.timing()
    part = H.lm_partials_device(s, J, base, market); t2 = time.perf_counter()
    ws.U.copy_(U0)
    trial = s.compute_base_prices_multi_maturity(S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, m1, m2, m, THETA, pts, len(pts), gd, ws)
    t3 = time.perf_counter(); tb = s.timing()
    print("jacobian wall %.2f ms (setup %.2f sweep %.2f finish %.2f) | partials %.2f ms | trial wall %.2f ms (setup %.2f sweep %.2f finish %.2f)" % (
        (t1 - t0) * 1e3, tj["setup_ms"], tj["sweep_ms"], tj["finish_ms"], (t2 - t1) * 1e3, (t3 - t2) * 1e3, tb["setup_ms"], tb["sweep_ms"], tb["finish_ms"]))
