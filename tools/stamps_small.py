#!/usr/bin/env python3
"""Diagnostic (not product): phase shares of the small-grid LDS-resident kernel."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "_stamps"); os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libhadi.so")
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DHADI_STAMPS=%s" % os.environ.get("STAMP_LEVEL", "2"),
                       "-o", lib, os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc", "hadi_api.hip")])
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
nat.LIB_PATH = lib
import pde_based_heston_solver_gpu_accelerated_amd as H
import numpy as np, torch
n, m1, m2, N = 500, 50, 25, 20
strikes = [85.0] * n
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, strikes); U0 = g.call_payoff(strikes)
dev = torch.device("cuda:0"); gd = g.to(dev); U = torch.from_numpy(U0).to(dev)
s = H.HestonADI(0)
L = nat.lib(); L.hadi_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
for _ in range(2):
    s.DO_timestepping(m1, m2, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
    L.hadi_debug_stamps(buf, 1)
names = {0: "row scalars + col0", 1: "LDS rows -> tt,A2U", 2: "coef + Y0 + fwd Thomas", 3: "bwd Thomas + reduced row",
         4: "PCR", 5: "final + store", 8: "dividend/col pass + loop top", 9: "barrier wait", 10: "row pass (per wave)"}
waves = n * 4 * N
print("sweep_ms", s.timing()["sweep_ms"])
for k, nm in names.items():
    print("%-32s %8.0f cycles/wave-step" % (nm, buf[k] / waves))
