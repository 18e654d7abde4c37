#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:
grid-points x timesteps / sec of the batched Douglas ADI sweep on a 512x256 grid, 1000 time steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--instances I]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: hadi_DO_timestepping on `I` independent
European calls per GPU (workload C2 of SURVEY.md 8(d): strikes 85..115, canonical Heston parameters),
inputs already resident in HBM, including operator setup, layout pack/unpack and all 1000 time steps.
Instances are sharded across ranks with no data-path collective (scaling: weak).

Extra objects on the JSON line:
  roofline      dominant kernel = the row pass (hadi_pass_a_strip at 8 nodes per lane, else hadi_pass_a).
                achieved = 16 B x points per launch / mean launch duration against the 8 TB/s HBM peak.  The mean
                launch duration comes from HIP events on the library's own stream: the time-loop events of the TIMED
                steps (hadi_get_timing().sweep_ms, 2 x timesteps launches) split between the two kernels by the
                per-launch events of one extra profiled step (hadi_set_profiling; `avg_launch_ms_profiled` is that
                step's own figure -- per-launch events cost a few percent).  `sweep` repeats it for the whole Douglas
                step (32 B per point-step, both passes).
  cpu_baseline  the CPU oracle (plain-C port of the reference's algorithm, OpenMP over instances) timed
                on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured float4 copy)
B_ALG_PASS = 16.0            # algorithmic bytes per grid point per pass (read once + write once, fp64)
B_ALG_STEP = 32.0            # per Douglas step: two directional passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=256, help="option instances per GPU")
    ap.add_argument("--m1", type=int, default=512)
    ap.add_argument("--m2", type=int, default=256)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--state", default="fp64", choices=["fp64", "fp32"],
                    help="precision of the state between the two passes (fp32 = BASELINE config 5: half the traffic, fp64 arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="skip the 1-instance latency run (profiling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a 1-GPU box cannot host one GPU per rank)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as G
    G.build()
    import pde_based_heston_solver_gpu_accelerated_amd as H
    global B_ALG_PASS, B_ALG_STEP
    STATE = H.STATE_FP32 if args.state == "fp32" else H.STATE_FP64
    if args.state == "fp32":  # 16 B per point-step with fp32 state
        B_ALG_PASS, B_ALG_STEP = 8.0, 16.0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev_index = 0 if args.share_device else local_rank
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(dev_index)
        collective = args.backend
        if args.backend == "nccl":
            # RCCL carries the only collectives of this bench (barrier + MAX of one double around the timed region; the data
            # path has none).  If RCCL cannot come up on this node the numbers are still worth having: every rank then
            # falls back to gloo for those two calls and the JSON line says so.
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
                probe = torch.ones(1, device=torch.device("cuda", dev_index))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                collective = "gloo (rccl unavailable: %s)" % str(exc).splitlines()[0][:120]
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                dist.init_process_group(backend="gloo")
                args.backend = "gloo"
        else:
            dist.init_process_group(backend=args.backend)
    else:
        collective = None
    n_gpus = world if world > 1 else 1
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    # ---- workload C2: this rank's shard of the strike ladder ---------------------------------------
    S_0, V_0, T, r_d, r_f = 100.0, 0.04, 1.0, 0.025, 0.0
    rho, sigma, kappa, eta, theta = -0.9, 0.3, 1.5, 0.04, 0.8
    m1, m2, N, n_loc = args.m1, args.m2, args.timesteps, args.instances
    n_glob = n_loc * n_gpus
    all_strikes = [100.0] if n_glob == 1 else [85.0 + 30.0 * k / (n_glob - 1) for k in range(n_glob)]
    strikes = all_strikes[rank * n_loc:(rank + 1) * n_loc]
    grids_h = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, strikes)
    U0_h = grids_h.call_payoff(strikes)
    grids = grids_h.to(dev)
    U0 = torch.from_numpy(U0_h).to(dev)
    U = torch.empty_like(U0)
    m = (m1 + 1) * (m2 + 1)
    solver = H.HestonADI(dev_index)

    def step():
        U.copy_(U0)  # workspace.U <- U_0 before every call (heston_calibration.cpp:216); D2D, part of the pass
        torch.cuda.synchronize()
        solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U,
                               state_precision=STATE)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    sweep_ms = 0.0
    for _ in range(args.steps):
        step()
        sweep_ms += solver.timing()["sweep_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    units = float(n_glob) * m * N * args.steps
    value = units / elapsed

    # price sanity inside the bench: every rank checks its instance nearest K = 100 is a sane call price
    k_mid = min(range(n_loc), key=lambda k: abs(strikes[k] - 100.0))
    g = H.Grid(m1, 8 * strikes[k_mid], S_0, strikes[k_mid], strikes[k_mid] / 5, m2, 5.0, V_0, 5.0 / 500)
    price_mid = float(U[k_mid, g.find_s_index(S_0) + g.find_v0_index(V_0) * (m1 + 1)].item())

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel, one extra profiled pass outside the timed region -----
        solver.set_profiling(True)
        step()
        tm = solver.timing()
        path = solver.describe_last_sweep()
        solver.set_profiling(False)
        pts = float(n_loc) * m
        a_prof = tm["pass_a_ms"] / max(1, tm["pass_a_launches"])
        b_prof = tm["pass_b_ms"] / max(1, tm["pass_b_launches"])
        # timed region: time-loop events of the K timed steps (same stream), split by the profiled step's shares
        loop_ms = sweep_ms / max(1, args.steps) / N
        share_a = tm["pass_a_ms"] / max(1e-30, tm["pass_a_ms"] + tm["pass_b_ms"])
        a_ms, b_ms = loop_ms * share_a, loop_ms * (1.0 - share_a)
        ach_a = B_ALG_PASS * pts / (a_ms * 1e-3) / 1e9
        ach_b = B_ALG_PASS * pts / (b_ms * 1e-3) / 1e9
        ach_step = B_ALG_STEP * pts / (loop_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": path.split(";")[0].replace("row pass ", ""), "kernels": path,
            "achieved": round(ach_a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_a / HBM_PEAK_GBS, 4),
            "traffic": None,
            "avg_launch_ms": round(a_ms, 5), "avg_launch_ms_profiled": round(a_prof, 5),
            "bytes_per_launch_algorithmic": B_ALG_PASS * pts,
            "pass_b": {"achieved": round(ach_b, 1), "frac": round(ach_b / HBM_PEAK_GBS, 4), "avg_launch_ms": round(b_ms, 5),
                       "avg_launch_ms_profiled": round(b_prof, 5)},
            "sweep": {"achieved": round(ach_step, 1), "frac": round(ach_step / HBM_PEAK_GBS, 4),
                      "bytes_per_point_step": B_ALG_STEP, "sweep_ms": round(sweep_ms / max(1, args.steps), 3),
                      "sweep_ms_profiled": round(tm["sweep_ms"], 3),
                      "setup_ms": round(tm["setup_ms"], 3), "finish_ms": round(tm["finish_ms"], 3)},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                key = "%dx%dx%d" % (m1, m2, n_loc)
                if key in rec:
                    roofline["traffic"] = rec[key]["pass_a_bytes_per_launch"]
                    roofline["traffic_source"] = rec[key].get("source", "profiles/pmc_traffic.json")
            except Exception:
                pass

        # ---- single-instance latency (the literal config[1]: ONE European call) ----------------------
        g1 = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, [100.0])
        u1 = g1.call_payoff([100.0])
        g1d = g1.to(dev)
        u1d = torch.from_numpy(u1).to(dev)
        w1 = torch.empty_like(u1d)
        best = 1e30
        for _ in range(0 if args.skip_single else 3):
            w1.copy_(u1d)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            solver.DO_timestepping(m1, m2, N, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, g1d, w1)
            best = min(best, time.perf_counter() - t1)
        gi = H.Grid(m1, 800.0, S_0, 100.0, 20.0, m2, 5.0, V_0, 5.0 / 500)
        price1 = float(w1[0, gi.find_s_index(S_0) + gi.find_v0_index(V_0) * (m1 + 1)].item()) if not args.skip_single else float("nan")

        # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores ------------
        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            cores = O.max_threads()
            n_cpu = max(1, min(cores, 64))
            N_cpu = max(10, min(N, 600))  # ~10-15 s wall on 64 cores (a bounded sample; the full workload would take ~80 s)
            gs = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, all_strikes[:n_cpu] if n_glob >= n_cpu else [100.0] * n_cpu)
            u0c = gs.call_payoff(all_strikes[:n_cpu] if n_glob >= n_cpu else [100.0] * n_cpu)
            p = O.make_params(m1, m2, N_cpu, T / N, theta, r_d, r_f, rho, sigma, kappa, eta, O.EU)
            t1 = time.perf_counter()
            _, _, used = O.solve_batch(p, gs.Vec_s, gs.Vec_v, gs.Delta_s, gs.Delta_v, u0c, threads=cores)
            dt_cpu = time.perf_counter() - t1
            cpu = {"value": n_cpu * m * N_cpu / dt_cpu, "unit": "point-steps/s", "cores": int(used), "kind": "port",
                   "sample": "%d instances x %d of the %d time steps of the same %dx%d workload, OpenMP over instances, %.1f s wall"
                             % (n_cpu, N_cpu, N, m1, m2, dt_cpu)}

        info = solver.device_info()
        out = {
            "metric": "grid-points x timesteps/sec (ADI sweep), %dx%d grid" % (m1, m2),
            "value": value, "unit": "point-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.state == "fp64" else "f32 state, f64 arithmetic", "data": "synthetic",
            "config": {"workload": "C2: European calls, Heston Douglas ADI, %dx%d grid, %d time steps, %d strikes per GPU "
                                   "(85..115), HBM-resident inputs" % (m1, m2, N, n_loc),
                       "m1": m1, "m2": m2, "timesteps": N, "instances_per_gpu": n_loc, "instances_total": n_glob,
                       "parallelism": "instances sharded over %d GPU(s), no data-path collective" % n_gpus,
                       "timing_collective": collective},
            "effective_GBps": value * B_ALG_STEP / 1e9,
            "sweep_only_point_steps_per_s": float(n_loc) * m * N * args.steps / (sweep_ms * 1e-3),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "single_instance": None if args.skip_single else {
                "wall_ms": best * 1e3, "price": price1, "reference_price": 8.8942192888223310,
                "price_abs_err": abs(price1 - 8.8942192888223310)},
            "price_check": {"strike": strikes[k_mid], "price": price_mid},
            "device": info,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    solver.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
