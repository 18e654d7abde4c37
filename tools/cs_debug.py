#!/usr/bin/env python3
"""Diagnostic (needs a GPU): Craig-Sneyd on strips, product build against the full-drain build, mode by mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pde_based_heston_solver_gpu_accelerated_amd as H
import __graft_entry__ as G
import common as Cm
m1, m2 = 512, 256
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prod, strict = H.HestonADI(0), H.HestonADI(0, lib_path=G.build_libhadi_strict())
strikes = Cm.strikes_for(n)
grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, strikes)
U0 = grids.call_payoff(strikes)
for N in (1, 2, 10, 60):
    args = (m1, m2, N, Cm.T / N, Cm.THETA, Cm.R_D, 0.01, Cm.RHO, Cm.SIGMA, Cm.KAPPA, Cm.ETA, grids)
    for key in (1, 2, 3, 0):
        res = []
        for sv in (prod, strict):
            sv.set_tuning("cs_strips", key)
            U = U0.copy()
            sv.CS_scheme(*args, U)
            res.append(U)
            sv.set_tuning("cs_strips", 1)
        d = np.abs(res[0] - res[1])
        bad = np.argwhere(d > 0)
        msg = ""
        if len(bad):
            inst = np.unique(bad[:, 0]); rows = np.unique((bad[:, 1] // (m1 + 1)))
            msg = " instances %s%s rows %s%s" % (inst[:8], "..." if len(inst) > 8 else "", rows[:12], "..." if len(rows) > 12 else "")
        print("N=%d cs_strips=%d (%s): max |product - strict| = %.3e, %d entries differ%s" % (
            N, key, {0: "ring", 1: "both on strips", 2: "predictor only", 3: "corrector only"}[key], d.max(), len(bad), msg), flush=True)
