#!/usr/bin/env python3
"""Times the other BASELINE.json configurations (parity-test cases, not the headline bench line) on one GPU:
   C3  batch of 512 American calls with discrete dividends, 256x128 grid, 500 steps
   C4  Heston LM calibration pieces on a 500-option surface (50 strikes x 10 maturities), 50x25 grid:
       one flattened Jacobian (3000 instances) + one trial pricing (500 instances), per-instance maturities
   C5  the config-5 grid (1024x512, 2000 steps), 64 instances: fp64, and with the state kept in fp32 between the passes
       (fp64 arithmetic) -- price difference reported
   REF the reference's own perf-harness shape: 50x25, N=20, 500 instances (perfomance_test.cpp:46-57)
Prints one JSON object; `python tools/bench_configs.py > profiles/<tag>_configs.json` on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd as H

S_0, V_0, T, r_d, r_f = 100.0, 0.04, 1.0, 0.025, 0.0
rho, sigma, kappa, eta, theta = -0.9, 0.3, 1.5, 0.04, 0.8
dev = torch.device("cuda:0")
solver = H.HestonADI(0)
out = {}


def ladder(n):
    return [100.0] if n == 1 else [85.0 + 30.0 * k / (n - 1) for k in range(n)]


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


# ---- C3
m1, m2, N, n = 256, 128, 500, 512
ks = ladder(n); g = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks); U0h = g.call_payoff(ks)
gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev); U = torch.empty_like(U0)
div = H.Dividends([0.2, 0.4, 0.6, 0.8], [0.5, 0.3, 0.2, 0.1], [0.02] * 4)
def c3():
    U.copy_(U0)
    solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, gd, U, variant=H.AM_DIV, U_0=U0, dividends=div)
t = timed(c3)
k100 = min(range(n), key=lambda k: abs(ks[k] - 100.0))
out["C3_american_dividend_256x128x500_x512"] = {"seconds": t, "point_steps_per_s": n * (m1 + 1) * (m2 + 1) * N / t,
                                                "effective_GBps_at_48B": n * (m1 + 1) * (m2 + 1) * N * 48 / t / 1e9,
                                                "sweep_ms": solver.timing()["sweep_ms"]}
# single K=100 instance of the same config: reference output 5.2760823084423789
g1 = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, [100.0]); u1 = g1.call_payoff([100.0])
ws = H.DOWorkspace(1, (m1 + 1) * (m2 + 1)); ws.U[...] = u1
p = solver.compute_base_prices_american_dividends(S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, (m1 + 1) * (m2 + 1), N,
                                                  theta, T / N, 1, g1, u1, ws, div)
out["C3_american_dividend_256x128x500_x512"]["price_K100"] = float(p[0])
out["C3_american_dividend_256x128x500_x512"]["price_abs_err_vs_reference"] = abs(float(p[0]) - 5.2760823084423789)
# ---- C3 with the put-shaped payoff of the literal config (no reference put path: algorithm parity only, see grid.put_payoff)
U0p = torch.from_numpy(g.put_payoff(ks)).to(dev)
def c3p():
    U.copy_(U0p)
    solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, gd, U, variant=H.AM_DIV, U_0=U0p, dividends=div)
t = timed(c3p)
out["C3_put_payoff_american_dividend_256x128x500_x512"] = {"seconds": t, "point_steps_per_s": n * (m1 + 1) * (m2 + 1) * N / t,
                                                            "kernels": solver.describe_last_sweep()}

del U, U0, U0p, gd

# ---- C4: 50 strikes x 10 maturities, N_m = max(20, floor(20 T_m)) (heston_calibration.cpp:2485-2517)
m1, m2 = 50, 25
strikes = [80.0 + 40.0 * k / 49 for k in range(50)]
mats = [0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0]
Ks = [K for Tm in mats for K in strikes]
Ts = [Tm for Tm in mats for K in strikes]
Ns = [max(20, int(20 * Tm)) for Tm in Ts]
g = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, Ks); U0h = g.call_payoff(Ks)
gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev)
per = {"N_i": Ns, "delta_t_i": [a / b for a, b in zip(Ts, Ns)]}
ts = (m1 + 1) * (m2 + 1)
def c4_jac():
    solver.compute_jacobian(S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, ts, 20, theta, T / 20, len(Ks), gd, U0, per_instance=per)
ws = H.DOWorkspace(len(Ks), ts, device=dev)
def c4_trial():
    ws.U.copy_(U0)
    solver.compute_base_prices(S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, ts, 20, theta, T / 20, len(Ks), gd, ws, per_instance=per)
tj, tt = timed(c4_jac), timed(c4_trial)
work = sum(ts * nn for nn in Ns)
out["C4_lm_iteration_500_options_50x25"] = {"jacobian_seconds": tj, "trial_seconds": tt, "pde_solves": 7 * len(Ks),
                                            "point_steps_per_s": 7 * work / (tj + tt), "seconds_per_lm_iteration": tj + tt}

# ---- reference perf-harness shape
n, N = 500, 20
ks = [85.0] * n; g = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks); U0h = g.call_payoff(ks)
gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev); U = torch.empty_like(U0)
def ref():
    U.copy_(U0)
    solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, gd, U)
t = timed(ref, 5)
out["REF_harness_50x25x20_x500_european"] = {"seconds": t, "point_steps_per_s": n * ts * N / t,
                                            "readme_a100_500_american_dividend_options_seconds": 0.02}
def ref_amdiv():
    U.copy_(U0)
    solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, gd, U, variant=H.AM_DIV, U_0=U0, dividends=div)
t = timed(ref_amdiv, 5)
out["REF_harness_50x25x20_x500_american_dividend"] = {"seconds": t, "point_steps_per_s": n * ts * N / t}

# ---- C5 grid: fp64 and fp32 state
m1, m2, N, n = 1024, 512, 2000, 64
ks = ladder(n); g = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks); U0h = g.call_payoff(ks)
gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev); U = torch.empty_like(U0)
prec = H.STATE_FP64
def c5():
    U.copy_(U0)
    solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, gd, U, state_precision=prec)
kmid = min(range(n), key=lambda k: abs(ks[k] - 100.0))
gk = H.Grid(m1, 8 * ks[kmid], S_0, ks[kmid], ks[kmid] / 5, m2, 5.0, V_0, 5.0 / 500)
node = gk.find_s_index(S_0) + gk.find_v0_index(V_0) * (m1 + 1)
t = timed(c5, 1)
p64 = float(U[kmid, node].item())
prec = H.STATE_FP32
t32 = timed(c5, 1)
p32 = float(U[kmid, node].item())
out["C5_european_1024x512x2000_x64_fp32_state"] = {"seconds": t32, "point_steps_per_s": n * (m1 + 1) * (m2 + 1) * N / t32,
                                                   "effective_GBps_at_16B": n * (m1 + 1) * (m2 + 1) * N / t32 * 16 / 1e9,
                                                   "price_fp32_state": p32, "price_fp64": p64, "price_abs_diff": abs(p32 - p64),
                                                   "kernels": solver.describe_last_sweep()}
out["C5grid_european_1024x512x2000_x64_fp64"] = {"seconds": t, "point_steps_per_s": n * (m1 + 1) * (m2 + 1) * N / t}
out["device"] = solver.device_info()
print(json.dumps(out, indent=1))
