#!/usr/bin/env python3
"""Diagnostic (needs a GPU): is the column pass's run-to-run spread (102 - 111 us at 512x256 x256, 112 - 119 us at 1024x512 x64: the
row pass of the same runs is steady to 1 %) a property of the PROCESS (where its allocations landed) or of the moment?
    python tools/colpass_modes.py [c2|c5] [handles] [repeats]
Creates `handles` solver handles one after the other in ONE process (each allocates its own scratch) and times the two passes
`repeats` times on each (per-launch events of a profiled sweep)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
handles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m1, m2, n, N = (512, 256, 256, 30) if wl == "c2" else (1024, 512, 64, 20)
ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks)
dev = torch.device("cuda:0")
u0 = torch.from_numpy(g.call_payoff(ks)).to(dev)
gd = g.to(dev)
args = (m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd)
keep = []
for h in range(handles):
    s = H.HestonADI(0)
    s.set_profiling(True)
    out = []
    for r in range(reps):
        u = u0.clone()
        torch.cuda.synchronize()
        s.DO_timestepping(*args, u)
        torch.cuda.synchronize()
        t = s.timing()
        out.append((1e3 * t["pass_a_ms"] / max(1, t["pass_a_launches"]), 1e3 * t["pass_b_ms"] / max(1, t["pass_b_launches"])))
    print("%s handle %d: " % (wl, h) + "  ".join("row %.1f col %.1f us" % o for o in out), flush=True)
    if h % 2 == 0:
        keep.append(s)      # (every other handle stays alive: the next one's scratch lands somewhere else)
    else:
        s.close()
