#!/usr/bin/env python3
"""Diagnostic (needs a GPU): Craig-Sneyd against Douglas on the headline geometry.
    python tools/cs_bench.py [instances] [steps]
Prints ms per time step of both schemes (device-resident inputs, sweep-only events) and the kernels chosen."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m1, m2 = 512, 256
ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks)
dev = torch.device("cuda:0")
u0 = torch.from_numpy(g.call_payoff(ks)).to(dev)
gd = g.to(dev)
s = H.HestonADI(0)
args = (m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd)
for name, fn in (("Douglas", lambda u: s.DO_timestepping(*args, u)), ("Craig-Sneyd", lambda u: s.CS_scheme(*args, u))):
    best = 1e30
    for rep in range(3):
        u = u0.clone()
        torch.cuda.synchronize()
        fn(u)
        torch.cuda.synchronize()
        best = min(best, s.timing()["sweep_ms"])
    pts = n * (m1 + 1) * (m2 + 1)
    print("%-12s %.4f ms per step  %.3e point-steps/s | %s" % (name, best / N, pts * N / (best * 1e-3), s.describe_last_sweep()), flush=True)
