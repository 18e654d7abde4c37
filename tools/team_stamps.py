#!/usr/bin/env python3
"""Diagnostic (GPU): phase stamps of one time step of hadi_team_kernel from a -DHADI_TEAM_STAMPS=<step> build
(python tools/build_variant.py tstamps -DHADI_TEAM_STAMPS=500; python tools/team_stamps.py tools/_var_tstamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
nat.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H
dev = torch.device("cuda:0"); s = H.HestonADI(0)
m1, m2, N = 512, 256, 1000
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, [100.0]); U0 = torch.from_numpy(g.call_payoff([100.0])).to(dev); gd = g.to(dev)
s.set_tuning("team_launch", 1)
names = ["loop top", "row phase done", "barrier 1 passed", "column loads landed", "column solve done", "column stores issued", "barrier 2 passed"]
for rep in range(3):
    U = U0.clone(); torch.cuda.synchronize()
    s.DO_timestepping(m1, m2, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
    out = (C.c_ulonglong * 16)()
    assert s._lib.hadi_debug_team_stamps(s._h, out) == 0
    t = list(out)
    print("sweep %.2f ms | %s" % (s.timing()["sweep_ms"], s.describe_last_sweep()[:40]))
    for base, who in ((0, "block 0 (column tile 0)"), (8, "last block of the team (rows only)")):
        t0 = t[base]
        print("  %s:" % who, ", ".join("%s +%d" % (names[k], t[base + k] - t0) for k in range(1, 7) if t[base + k] >= t0 and t[base + k] != 0))
