#!/usr/bin/env python3
"""Diagnostic (not product): builds libhadi with -DHADI_STAMPS into tools/_stamps/ and prints where a
row-pass wavefront spends its cycles (shares, not absolute speed -- the stamps serialise the phases)."""
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
n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
m1, m2 = int(os.environ.get("M1", "512")), int(os.environ.get("M2", "256"))
strikes = [85 + 30 * k / max(1, n - 1) for k in range(n)]
g = H.GridViewsBatch.for_strikes(m1, m2, 100.0, 0.04, strikes); U0 = g.call_payoff(strikes)
dev = torch.device("cuda:0"); gd = g.to(dev); U = torch.from_numpy(U0).to(dev)
s = H.HestonADI(0)
L = nat.lib(); L.hadi_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
s.DO_timestepping(m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
L.hadi_debug_stamps(buf, 1)
s.DO_timestepping(m1, m2, N, 1.0 / 1000, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U)
L.hadi_debug_stamps(buf, 1)
names = {0: "row scalars + col0", 1: "LDS rows -> tt,A2U", 2: "coef + Y0 + fwd Thomas", 3: "bwd Thomas + reduced row",
         4: "PCR", 5: "final + store", 8: "prologue", 9: "barrier wait", 10: "row step total (+fetch issue)"}
if os.environ.get("STAMP_LEVEL") == "3":
    bn = {16: "wait for the tile's loads", 17: "forward", 18: "backward", 19: "exchange + barrier", 20: "reduced system",
          21: "spike correction", 22: "store issue"}
    P = (m2 + 1 + 32) // 33; BG = 8 * (2 if m1 > 512 else 1) if m1 > 256 else (4 if m1 > 128 else 2 if m1 > 64 else 1)
    tiles = n * P * (BG + 1) * N  # P chunks x column tiles (full + the short one)
    tb = sum(buf[k] for k in bn)
    print("sweep_ms", s.timing()["sweep_ms"])
    for k, nm in bn.items():
        print("%-32s %8.0f cycles/tile  %5.1f %%" % (nm, buf[k] / tiles, 100.0 * buf[k] / tb))
    sys.exit(0)
if os.environ.get("STAMP_LEVEL") == "4":
    cn = {24: "wait for the DMA (row j+2)", 31: "DMA issue (row j+4)", 25: "LDS reads + table entry", 26: "explicit ops + Y0 + fwd Thomas",
          27: "bwd Thomas + reduced row", 28: "PCR", 29: "final + store issue", 30: "carry + loop"}
    rows = n * (m2 + 1) * N
    tc = sum(buf[k] for k in cn)
    print("sweep_ms", s.timing()["sweep_ms"])
    for k, nm in cn.items():
        print("%-40s %8.0f cycles/row  %5.1f %%" % (nm, buf[k] / rows, 100.0 * buf[k] / tc))
    sys.exit(0)
rows = n * (m2 + 1) * N
tot = sum(buf[k] for k in (8, 9, 10))
print("sweep_ms", s.timing()["sweep_ms"])
for k, nm in names.items():
    print("%-32s %8.0f cycles/row  %5.1f %%" % (nm, buf[k] / rows, 100.0 * buf[k] / tot))
