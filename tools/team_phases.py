#!/usr/bin/env python3
"""Timing diagnostics of hadi_team_kernel (results of the debug runs are WRONG on purpose): the per-step time with the row
phase, the column phase or the team barriers switched off (hadi_set_tuning "debug_fault" bits 16 / 32 / 64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H
dev = torch.device("cuda:0"); s = H.HestonADI(0)
m1, m2, N = 512, 256, 1000
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ks = [100.0] if n == 1 else [85.0 + 30.0 * k / (n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks); U0 = torch.from_numpy(g.call_payoff(ks)).to(dev); gd = g.to(dev)
s.set_tuning("team_launch", 1)
for dbg, name in ((0, "all"), (16, "no rows"), (32, "no columns"), (48, "barriers only"), (64, "no barriers"), (112, "nothing"),
                  (16 + 128, "cols w/o loads"), (16 + 256, "cols w/o solve"), (16 + 512, "cols w/o stores"), (16 + 128 + 512, "cols: solve only"),
                  (16 + 256 + 512, "cols: loads only"), (16 + 128 + 256, "cols: stores only")):
    s.set_tuning("debug_fault", dbg)
    U = torch.empty_like(U0); best = 1e9
    for _ in range(3):
        U.copy_(U0); torch.cuda.synchronize()
        s.DO_timestepping(m1, m2, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
        best = min(best, s.timing()["sweep_ms"])
    print("%-14s %.2f us per step   (%s)" % (name, best * 1e3 / N, s.describe_last_sweep()[:30]), flush=True)
s.set_tuning("debug_fault", 0); s.set_tuning("team_launch", -1)
