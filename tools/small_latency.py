import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
if os.environ.get("HADI_LIB"): nat.LIB_PATH = os.path.abspath(os.environ["HADI_LIB"])
import pde_based_heston_solver_gpu_accelerated_amd as H
dev = torch.device("cuda:0"); s = H.HestonADI(0)
if os.environ.get("SMALL_SEQ"): s.set_tuning("small_seq", int(os.environ["SMALL_SEQ"]))
def run(n, N, variant=H.EU):
    ks = [85.0 + 30.0 * k / max(1, n - 1) for k in range(n)]
    g = H.GridViewsBatch.for_strikes(50, 25, 100.0, 0.04, ks); U0h = g.call_payoff(ks)
    gd = g.to(dev); U0 = torch.from_numpy(U0h).to(dev); U = torch.empty_like(U0)
    best = 1e9
    for _ in range(5):
        U.copy_(U0); torch.cuda.synchronize(); t = time.perf_counter()
        s.DO_timestepping(50, 25, N, 1.0 / N, 0.8, 0.025, 0.0, -0.9, 0.3, 1.5, 0.04, gd, U, variant=variant, U_0=U0)
        best = min(best, time.perf_counter() - t)
    return best * 1e3, float(U[0, 500].item())
cases = ((1, 100), (60, 20), (500, 20), (500, 100), (1024, 50), (3000, 50))
if os.environ.get("SIZES"): cases = tuple((int(x), int(os.environ.get("STEPS", "40"))) for x in os.environ["SIZES"].split(","))
for n, N in cases:
    print(n, N, "%.3f ms" % run(n, N)[0], run(n, N)[1])
