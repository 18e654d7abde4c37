#!/usr/bin/env python3
"""The 16-chunk column pass (hadi_pass_b1, BASELINE config 5) against its two alternatives of round 4 and against its own memory
traffic alone, one process, one box:   python tools/colpass_ab.py > profiles/r04_colpass_ab.txt
  b1            the default: one register buffer, the next tile loaded behind the stores
  b2            "col_prefetch": 12 (fp32 state: 24) of the 33 rows of the next tile prefetched into LDS by LDS-DMA before the solve
  +il           "tile_interleave": the 4 blocks of an instance walk the 16 full tiles interleaved instead of 4 consecutive ones each
  no solve      "debug_fault" 256 (results wrong): tiles loaded and stored, not solved -- the pass's access pattern alone
  no R^-1z      "debug_fault" 512 (results wrong): everything but the reduced system t = R^-1 z (4 x 4P broadcast-operand FMAs per lane)
Mean launch time of the column pass over a profiled sweep (per-launch HIP events on the library's stream), best of 3 sweeps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pde_based_heston_solver_gpu_accelerated_amd as H

S_0, V_0, T, R_D, R_F = 100.0, 0.04, 1.0, 0.025, 0.0
RHO, SIGMA, KAPPA, ETA, THETA = -0.9, 0.3, 1.5, 0.04, 0.8
dev = torch.device("cuda:0")
s = H.HestonADI(0)
s.set_profiling(True)
print("%-28s %-10s %10s %10s %8s" % ("case", "variant", "col ms", "row ms", "col frac"))
for tag, m1, m2, N, n, state in (("c5 1024x512 x64 fp64", 1024, 512, 60, 64, "fp64"), ("c5 1024x512 x64 fp32", 1024, 512, 60, 64, "fp32"),
                                 ("1024x512 x128 fp64", 1024, 512, 40, 128, "fp64"), ("700x300 x128 fp64", 700, 300, 60, 128, "fp64"),
                                 ("512x400 x160 fp64", 512, 400, 60, 160, "fp64"), ("c2 512x256 x256 fp64", 512, 256, 100, 256, "fp64"),
                                 ("c3-size 256x128 x512 EU", 256, 128, 100, 512, "fp64")):
    ks = [85.0 + 30.0 * k / (n - 1) for k in range(n)]
    gh = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks)
    u0 = torch.from_numpy(gh.call_payoff(ks)).to(dev)
    gd, u = gh.to(dev), torch.empty_like(u0)
    by = 8.0 if state == "fp32" else 16.0
    for name, pf, il, dbg in (("b1", 0, 0, 0), ("b1 +il", 0, 1, 0), ("b2", 1, 0, 0), ("b2 +il", 1, 1, 0), ("b1 no solve", 0, 0, 256), ("b2 no solve", 1, 0, 256), ("b1 no R^-1z", 0, 0, 512)):
        s.set_tuning("col_prefetch", pf); s.set_tuning("tile_interleave", il); s.set_tuning("debug_fault", dbg)
        best = (1e30, 0.0)
        for rep in range(4):
            u.copy_(u0)
            s.DO_timestepping(m1, m2, N, T / 2000, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, gd, u,
                              state_precision=H.STATE_FP32 if state == "fp32" else H.STATE_FP64)
            tm = s.timing()
            if rep:
                best = min(best, (tm["pass_b_ms"] / tm["pass_b_launches"], tm["pass_a_ms"] / tm["pass_a_launches"]))
        frac = by * n * (m1 + 1) * (m2 + 1) / (best[0] * 1e-3) / 8e12
        print("%-28s %-10s %10.5f %10.5f %8.3f" % (tag, name, best[0], best[1], frac), flush=True)
    s.set_tuning("col_prefetch", 0); s.set_tuning("tile_interleave", 0); s.set_tuning("debug_fault", 0)
    del gd, u, u0
