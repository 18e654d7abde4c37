#!/usr/bin/env python3
"""Throughput of the sequential passes (grids beyond the streaming kernels: m1 > 1024 -> hadi_pass_a_seq, m2 > 527 ->
hadi_pass_b_seq), per-launch means from a profiled sweep:   python tools/seq_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pde_based_heston_solver_gpu_accelerated_amd as H

dev = torch.device("cuda:0")
s = H.HestonADI(0)
s.set_profiling(True)
print("%-22s %10s %10s %12s %9s | kernels" % ("shape x instances", "row ms", "col ms", "pt-steps/s", "of 32 B"))
for m1, m2, n, N in ((1500, 600, 8, 20), (1500, 600, 64, 10), (300, 800, 8, 20), (300, 800, 128, 10), (2100, 40, 64, 20), (1030, 530, 32, 10)):
    ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
    gh = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, ks)
    u0 = torch.from_numpy(gh.call_payoff(ks)).to(dev)
    gd, u = gh.to(dev), torch.empty_like(u0)
    best = None
    for rep in range(3):
        u.copy_(u0)
        s.DO_timestepping(m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, u)
        tm = s.timing()
        cur = (tm["sweep_ms"], tm["pass_a_ms"] / tm["pass_a_launches"], tm["pass_b_ms"] / tm["pass_b_launches"])
        if rep and (best is None or cur[0] < best[0]):
            best = cur
    pts = n * (m1 + 1) * (m2 + 1)
    rate = pts * N / (best[0] * 1e-3)
    print("%-22s %10.4f %10.4f %12.3e %9.3f | %s" % ("%dx%d x%d" % (m1, m2, n), best[1], best[2], rate, rate * 32 / 8e12, s.describe_last_sweep()[:110]), flush=True)
    del gd, u, u0
