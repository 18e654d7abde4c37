#!/usr/bin/env python3
"""Diagnostic (needs a GPU): Craig-Sneyd against Douglas on the headline geometry.
    python tools/cs_bench.py [instances] [steps] [m1] [m2]
Prints ms per time step of both schemes (device-resident inputs, sweep-only events), the fraction of the 8 TB/s roofline at
32 B (Douglas) / 96 B (Craig-Sneyd) per point and step, and the kernels chosen; Craig-Sneyd on the strips (round 4) and on
the shared ring (tuning key cs_strips = 0: the path of rounds 2 - 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m1 = int(sys.argv[3]) if len(sys.argv) > 3 else 512
m2 = int(sys.argv[4]) if len(sys.argv) > 4 else 256
ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks)
dev = torch.device("cuda:0")
u0 = torch.from_numpy(g.call_payoff(ks)).to(dev)
gd = g.to(dev)
s = H.HestonADI(0)
args = (m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd)
def cs_ring(u):
    s.set_tuning("cs_strips", 0)
    try:
        s.CS_scheme(*args, u)
    finally:
        s.set_tuning("cs_strips", 1)
for name, fn, nbytes in (("Douglas", lambda u: s.DO_timestepping(*args, u), 32), ("Craig-Sneyd", lambda u: s.CS_scheme(*args, u), 96),
                         ("CS, shared ring", cs_ring, 96)):
    best = 1e30
    for rep in range(3):
        u = u0.clone()
        torch.cuda.synchronize()
        fn(u)
        torch.cuda.synchronize()
        best = min(best, s.timing()["sweep_ms"])
    pts = n * (m1 + 1) * (m2 + 1)
    print("%dx%d x%d %-16s %.4f ms per step  %.3e point-steps/s  %.3f of the roofline at %d B | %s"
          % (m1, m2, n, name, best / N, pts * N / (best * 1e-3), pts * N * nbytes / (best * 1e-3) / 8e12, nbytes, s.describe_last_sweep()), flush=True)
