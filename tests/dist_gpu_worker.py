"""Worker of tests/test_distributed_gpu.py: one rank of a 2-rank multi-maturity calibration with the GPU solver.
Launched with torch.distributed.run; both ranks share GPU 0 (a 1-GPU box), the collective runs over gloo -- on a real
node every rank has its own GPU and the backend is nccl (= RCCL), see bench.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import pde_based_heston_solver_gpu_accelerated_amd as H
import common as Cm


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m1, m2 = 50, 25
    mats = [1.0, 1.25, 1.5, 2.0, 3.0]
    strikes = [95.0 + 1.0 * i for i in range(12)]
    pts = H.make_calibration_points(strikes, mats)
    costs = [p.time_steps for p in pts]                      # balance by work: N_m grows with the maturity
    lo, hi = H.shard_range(len(pts), world, rank, costs)
    mine = pts[lo:hi]
    ks = [p.strike for p in mine]
    grids = H.GridViewsBatch.for_strikes(m1, m2, Cm.S_0, Cm.V_0, ks)
    U0 = grids.call_payoff(ks)
    market = np.array([H.market.call_price(Cm.S_0, p.strike, Cm.R_D, 0.2, p.maturity) for p in mine])
    solver = H.HestonADI(0)
    if "--small-seq" in sys.argv:  # pin the small-grid kernel whatever the shard size (the automatic choice goes by batch size)
        solver.set_tuning("small_seq", int(sys.argv[sys.argv.index("--small-seq") + 1]))
    if "--device-arrays" in sys.argv:
        # HBM-resident shard: Jacobian rows and prices stay on the GPU, hadi_lm_partials_device reduces them there and only
        # the 31 doubles of the all-reduce leave it
        dev = torch.device("cuda", 0)
        grids, U0 = grids.to(dev), torch.from_numpy(U0).to(dev)
    comm = H.Communicator(group=dist.new_group(backend="gloo") if "--subgroup" in sys.argv else None)
    res = H.calibrate_european_multi_maturity(solver, Cm.S_0, Cm.R_D, Cm.R_F, Cm.KAPPA, Cm.ETA, Cm.SIGMA, Cm.RHO, Cm.V_0, m1, m2,
                                              Cm.THETA, mine, grids, U0, market, comm=comm, n_total=len(pts), max_iter=6)
    counts = [H.shard_range(len(pts), world, r, costs)[1] - H.shard_range(len(pts), world, r, costs)[0] for r in range(world)]
    prices = comm.allgather_rows(res["model_prices"], counts)
    if rank == 0:
        out = {k: float(res[k]) for k in ("kappa", "eta", "sigma", "rho", "v0", "final_error")}
        out.update(iterations=res["iterations"], pde_solves=res["pde_solves"], prices=prices.tolist(), world=world,
                   shard=[lo, hi], errors=[h["error"] for h in res["history"]])
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
