#!/usr/bin/env python3
"""Diagnostic: per-kernel time split of BASELINE config 3 (512 American calls with dividends, 256x128x500)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
if os.environ.get("HADI_LIB"): nat.LIB_PATH = os.path.abspath(os.environ["HADI_LIB"])  # before anything loads the library
import pde_based_heston_solver_gpu_accelerated_amd as H
variant = {"EU": H.EU, "AM": H.AM, "DIV": H.DIV, "AM_DIV": H.AM_DIV}[sys.argv[1] if len(sys.argv) > 1 else "AM_DIV"]
m1, m2, N, n = [int(x) for x in os.environ.get("C3_SHAPE", "256,128,500,512").split(",")]
ks = [85.0 + 30.0 * k / (n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks); U0h = g.call_payoff(ks)
dev = torch.device("cuda:0"); gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev); U = torch.empty_like(U0)
div = H.Dividends([0.2, 0.4, 0.6, 0.8], [0.5, 0.3, 0.2, 0.1], [0.02] * 4) if variant in (H.DIV, H.AM_DIV) else None
s = H.HestonADI(0)
for prof in (False, True):
    s.set_profiling(prof)
    for _ in range(2):
        U.copy_(U0)
        s.DO_timestepping(m1, m2, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U, variant=variant, U_0=U0, dividends=div)
    t = s.timing()
    print("profiling" if prof else "plain", {k: round(v, 4) if isinstance(v, float) else v for k, v in t.items()})
print(s.describe_last_sweep())
pts = n * (m1 + 1) * (m2 + 1)
print("row pass ms/launch %.4f  column pass ms/launch %.4f  point-steps/s %.3e" % (t["pass_a_ms"] / N, t["pass_b_ms"] / N, pts * N / (t["sweep_ms"] * 1e-3)))
