#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:
grid-points x timesteps / sec of the batched Douglas ADI sweep (512x256 grid, 1000 time steps by default).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--instances I]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch, inputs already resident in HBM, including operator setup, layout
pack/unpack and all time steps.  Workloads (BASELINE.json configs 2..5):
  c2  (default, the metric's config) hadi_DO_timestepping on I European calls per GPU, 512x256 grid, 1000 steps
  c3  I = 512 American PUTS with discrete dividends per GPU, 256x128 grid, 500 steps (HADI_PUT boundary data)
  c5  I = 64 European calls per GPU, 1024x512 grid, 2000 steps, fp32 state between the passes (fp64 arithmetic)
  c4  one Levenberg-Marquardt iteration on a 500-option surface (50 strikes x 10 maturities, 50x25 grid): flattened
      Jacobian sweep (3000 solves) + device-side J^T J reduction + all-reduce of 31 doubles + trial prices (500 solves);
      the options are sharded over the ranks by cost, the all-reduce is INSIDE the timed region
Instances are sharded across ranks with no data-path collective (scaling: weak; c4: strong -- the surface is fixed).
Collectives of the timed region (barrier, MAX of the elapsed time, c4's LM all-reduce) run over RCCL; if RCCL cannot
come up on every rank the run exits non-zero unless --allow-gloo is given (rehearsals on a box without one GPU per rank).

Extra objects on the JSON line:
  roofline      dominant kernel = the row pass.  achieved = algorithmic bytes per launch / mean launch duration against
                the 8 TB/s HBM peak; the mean launch duration comes from HIP events on the library's own stream: the
                time-loop events of the TIMED steps (hadi_get_timing().sweep_ms) split between the two kernels by the
                per-launch events of one extra profiled step (hadi_set_profiling).  `sweep` repeats it for the whole
                Douglas step.  Algorithmic bytes per point: fp64 European 16 + 16; fp32 state 8 + 8; American in the
                P representation 16 (row pass) + 24 (column pass: Y, P_old in, P out) = 40 -- what this algorithm has to
                move; SURVEY.md 8(d) budgets 48 for an explicit lambda_bar array.
  batch_sweep   (c2, one GPU) the same workload at 64 / 160 / 192 / 256 / 512 instances
  cpu_baseline  the CPU oracle (plain-C port of the reference's algorithm, OpenMP over instances) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured float4 copy)

S_0, V_0, T, R_D, R_F = 100.0, 0.04, 1.0, 0.025, 0.0
RHO, SIGMA, KAPPA, ETA, THETA = -0.9, 0.3, 1.5, 0.04, 0.8
DIVS = ([0.2, 0.4, 0.6, 0.8], [0.5, 0.3, 0.2, 0.1], [0.02] * 4)

WORKLOADS = {
    #       m1    m2    N     instances/GPU  bytes pass A, pass B
    "c2": (512, 256, 1000, 256, 16.0, 16.0),
    "c3": (256, 128, 500, 512, 16.0, 24.0),
    "c5": (1024, 512, 2000, 64, 8.0, 8.0),
    "c4": (50, 25, 0, 500, 16.0, 16.0),
}


def init_distributed(args, torch, dev_index):
    """Default group: gloo (rendezvous + the RCCL vote).  Timed-region collectives: an RCCL group, used only if EVERY
    rank brought it up and passed a probe all-reduce; otherwise all ranks leave together (non-zero) or, with
    --allow-gloo, all fall back together."""
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
    if args.backend != "nccl":
        return dist, None, args.backend
    ok, why, grp = 1, "", None
    try:
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120),
                             device_id=torch.device("cuda", dev_index))
        probe = torch.ones(1, device=torch.device("cuda", dev_index))
        dist.all_reduce(probe, group=grp)
        torch.cuda.synchronize()
        ok = 1 if int(probe.item()) == dist.get_world_size() else 0
    except Exception as exc:  # noqa: BLE001
        ok, why = 0, str(exc).splitlines()[0][:160]
    vote = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)  # over gloo: every rank learns whether RCCL works everywhere
    if int(vote.item()) == 1:
        return dist, grp, "nccl"
    if not args.allow_gloo:
        if dist.get_rank() == 0:
            print("bench.py: RCCL did not come up on every rank (%s); refusing to report a multi-GPU number over gloo "
                  "(pass --allow-gloo for a rehearsal)" % (why or "a peer failed"), file=sys.stderr)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3)
    return dist, None, "gloo (rccl unavailable%s)" % (": " + why if why else "")


def surface_points(H):
    """50 strikes x 10 maturities, N_m = max(20, 20 T_m) (heston_calibration.cpp:2485-2531)."""
    mats = [1.0 + i * 0.25 if i < 8 else 3.0 + (i - 8) * 0.5 for i in range(10)]
    return H.make_calibration_points([S_0 * 0.75 + 1.0 * i for i in range(50)], mats)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: this process becomes the launcher.  It starts N copies of
    itself -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on
    127.0.0.1 -- BEFORE anything here touches the GPU or imports torch, relays rank 0's JSON line and exits non-zero if any
    rank failed (the same contract as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)."""
    import socket
    import subprocess
    import __graft_entry__ as G
    G.build()  # once, here: the ranks then find the artefacts up to date (hipcc needs no GPU)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    if any(codes):
        print("bench.py: rank exit codes %s" % codes, file=sys.stderr)
        sys.exit(next(c for c in codes if c) if all(isinstance(c, int) for c in codes) else 1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--instances", type=int, default=0, help="option instances per GPU (0 = the workload's own)")
    ap.add_argument("--m1", type=int, default=0)
    ap.add_argument("--m2", type=int, default=0)
    ap.add_argument("--timesteps", type=int, default=0)
    ap.add_argument("--state", default=None, choices=["fp64", "fp32"],
                    help="precision of the state between the two passes (c5 defaults to fp32: half the traffic, fp64 arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="skip the 1-instance latency run and the batch sweep (profiling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the timed collectives (nccl = RCCL)")
    ap.add_argument("--allow-gloo", action="store_true", help="rehearsal: fall back to gloo when RCCL cannot come up")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="hadi_set_tuning override (diagnostics), e.g. --tuning strip=1")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a 1-GPU box cannot host one GPU per rank)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            spawn_ranks(args, sys.argv[1:])  # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        ap.error("--gpus %d does not match the launcher's WORLD_SIZE=%s: n_gpus on the result line is always --gpus"
                 % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.state and args.workload in ("c3", "c4"):
        ap.error("--state applies to the European workloads c2 and c5 (c3 is American, c4 runs LDS-resident in fp64)")

    import numpy as np
    import torch
    import __graft_entry__ as G
    G.build()
    import pde_based_heston_solver_gpu_accelerated_amd as H

    wl = args.workload
    m1, m2, N, n_loc, BA, BB = WORKLOADS[wl]
    m1, m2, N = args.m1 or m1, args.m2 or m2, args.timesteps or N
    n_loc = args.instances or n_loc
    state = args.state or ("fp32" if wl == "c5" else "fp64")
    if wl in ("c2", "c5"):
        BA, BB = (8.0, 8.0) if state == "fp32" else (16.0, 16.0)
    STATE = H.STATE_FP32 if state == "fp32" else H.STATE_FP64
    B_STEP = BA + BB

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dist, group, collective = None, None, None
    if world > 1:
        dist, group, collective = init_distributed(args, torch, dev_index)
    n_gpus = args.gpus  # (== world: checked above)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if (group is not None) else torch.device("cpu")
    comm = H.Communicator(device=coll_dev, group=group)
    m = (m1 + 1) * (m2 + 1)
    solver = H.HestonADI(dev_index)
    for kv in args.tuning:
        solver.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))

    # ---- this rank's shard of the workload ---------------------------------------------------------------------------
    if wl == "c4":
        pts_all = surface_points(H)
        costs = [float(p.time_steps) for p in pts_all]
        lo, hi = H.shard_range(len(pts_all), n_gpus, rank, costs=costs)
        pts = pts_all[lo:hi]
        strikes = [p.strike for p in pts]
        grids_h = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, strikes)
        U0_h = grids_h.call_payoff(strikes)
        grids, U0 = grids_h.to(dev), torch.from_numpy(U0_h).to(dev)
        market = torch.tensor([H.market.call_price(S_0, p.strike, R_D, 0.2, p.maturity) for p in pts], dtype=torch.float64, device=dev)
        ws = H.DOWorkspace(len(pts), m, device=dev)
        units_step = 7.0 * m * sum(p.time_steps for p in pts_all)   # 6 Jacobian solves + 1 trial solve per option
        n_glob = len(pts_all)
        state_holder = {"price": float("nan")}

        def step():
            J, base = solver.compute_jacobian_multi_maturity(S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, m1, m2, m, THETA,
                                                             pts, len(pts), grids, U0)
            part = comm.allreduce_sum(H.lm_partials_device(solver, J, base, market))   # RCCL, inside the timed region
            delta = H.lm_solve(part, 0.01)
            new = H.clamp_parameters(KAPPA + delta[0], ETA + delta[1], SIGMA + delta[2], RHO + delta[3], V_0 + delta[4])
            ws.U.copy_(U0)
            trial = solver.compute_base_prices_multi_maturity(S_0, new[4], R_D, R_F, new[3], new[2], new[0], new[1], m1, m2, m,
                                                              THETA, pts, len(pts), grids, ws)
            solver.wait_stream()
            err = comm.allreduce_sum(np.array([float(((market - trial) ** 2).sum().item())]))
            state_holder["price"], state_holder["err"] = float(base[0].item()), float(err[0])
    else:
        n_glob = n_loc * n_gpus
        all_strikes = [100.0] if n_glob == 1 else [85.0 + 30.0 * k / (n_glob - 1) for k in range(n_glob)]
        strikes = all_strikes[rank * n_loc:(rank + 1) * n_loc]
        grids_h = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, strikes)
        U0_h = grids_h.put_payoff(strikes) if wl == "c3" else grids_h.call_payoff(strikes)
        grids, U0 = grids_h.to(dev), torch.from_numpy(U0_h).to(dev)
        U = torch.empty_like(U0)
        units_step = float(n_glob) * m * N
        kw = {}
        if wl == "c3":
            kw = dict(variant=H.AM_DIV, U_0=U0, dividends=H.Dividends(*DIVS), option_type=H.PUT, strikes=strikes)

        def step():
            U.copy_(U0)  # workspace.U <- U_0 before every call (heston_calibration.cpp:216); D2D, part of the pass.  No
            # synchronize: the launcher orders the library's stream after torch's (hadi_wait_stream)
            solver.DO_timestepping(m1, m2, N, T / N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, grids, U,
                                   state_precision=STATE, **kw)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if group is not None:
                dist.barrier(group=group)
            else:
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(tt.item())
    value = units_step * args.steps / elapsed

    out = None
    if rank == 0:
        info = solver.device_info()
        roofline, single, sweep_obj, price_check, cpu = None, None, None, None, None
        if wl != "c4":
            # price sanity inside the bench: the instance nearest K = 100
            k_mid = min(range(n_loc), key=lambda k: abs(strikes[k] - 100.0))
            g = H.Grid(m1, 8 * strikes[k_mid], S_0, strikes[k_mid], strikes[k_mid] / 5, m2, 5.0, V_0, 5.0 / 500)
            price_check = {"strike": strikes[k_mid],
                           "price": float(U[k_mid, g.find_s_index(S_0) + g.find_v0_index(V_0) * (m1 + 1)].item())}
            # ---- roofline of the dominant kernel, one extra profiled pass outside the timed region -----
            solver.set_profiling(True)
            step()
            tm = solver.timing()
            path = solver.describe_last_sweep()
            solver.set_profiling(False)
            pts_l = float(n_loc) * m
            a_prof = tm["pass_a_ms"] / max(1, tm["pass_a_launches"])
            b_prof = tm["pass_b_ms"] / max(1, tm["pass_b_launches"])
            loop_ms = sweep_ms / max(1, args.steps) / N   # time-loop events of the timed steps (incl. the rare dividend launches)
            share_a = tm["pass_a_ms"] / max(1e-30, tm["pass_a_ms"] + tm["pass_b_ms"])
            a_ms, b_ms = loop_ms * share_a, loop_ms * (1.0 - share_a)
            ach_a = BA * pts_l / (a_ms * 1e-3) / 1e9
            ach_b = BB * pts_l / (b_ms * 1e-3) / 1e9
            ach_step = B_STEP * pts_l / (loop_ms * 1e-3) / 1e9
            roofline = {
                "bound": "hbm", "kernel": path.split(";")[0].replace("row pass ", ""), "kernels": path,
                "achieved": round(ach_a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_a / HBM_PEAK_GBS, 4),
                "traffic": None,
                "avg_launch_ms": round(a_ms, 5), "avg_launch_ms_profiled": round(a_prof, 5),
                "bytes_per_launch_algorithmic": BA * pts_l, "bytes_per_point_algorithmic": BA,
                "pass_b": {"achieved": round(ach_b, 1), "frac": round(ach_b / HBM_PEAK_GBS, 4), "avg_launch_ms": round(b_ms, 5),
                           "avg_launch_ms_profiled": round(b_prof, 5), "bytes_per_point_algorithmic": BB},
                "sweep": {"achieved": round(ach_step, 1), "frac": round(ach_step / HBM_PEAK_GBS, 4),
                          "bytes_per_point_step": B_STEP, "sweep_ms": round(sweep_ms / max(1, args.steps), 3),
                          "sweep_ms_profiled": round(tm["sweep_ms"], 3),
                          "setup_ms": round(tm["setup_ms"], 3), "finish_ms": round(tm["finish_ms"], 3)},
            }
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    key = "%s%s:%dx%dx%d" % (wl, "f64" if (wl == "c5" and state == "fp64") else "", m1, m2, n_loc)
                    key = key if key in rec else "%dx%dx%d" % (m1, m2, n_loc) if wl == "c2" else key
                    if key in rec:
                        roofline["traffic"] = rec[key]["pass_a_bytes_per_launch"]
                        roofline["traffic_source"] = rec[key].get("source", "profiles/pmc_traffic.json")
                except Exception:  # noqa: BLE001
                    pass
        else:
            tm = solver.timing()
            path = solver.describe_last_sweep()
            roofline = {"bound": "hbm", "kernel": path, "achieved": round(value * 32.0 / 1e9 / n_gpus, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(value * 32.0 / 1e9 / n_gpus / HBM_PEAK_GBS, 4), "traffic": None,
                        "note": "LDS-resident kernel: the whole time loop of an instance runs out of LDS, HBM sees the state twice "
                                "per SOLVE; `achieved` is the EFFECTIVE rate at 32 B per point-step, not HBM traffic"}
            price_check = {"strike": strikes[0], "price": state_holder["price"], "trial_error": state_holder.get("err")}

        if wl == "c2" and n_gpus == 1 and not args.skip_single:
            # ---- the same workload at the batch sizes of SURVEY.md 8(d) C2 (one warm-up + one timed pass each) -----
            sweep_obj = {}
            # (160 and 192: batches that leave a partial round of CUs idle, also with hadi_set_tuning("streams", 2) -- two
            # halves side by side on two streams; an opt-in)
            for nb, streams in ((64, 1), (160, 1), (160, 2), (192, 1), (192, 2), (256, 1), (512, 1)):
                ks = [85.0 + 30.0 * k / (nb - 1) for k in range(nb)]
                solver.set_tuning("streams", streams)
                gb = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks)
                ub0 = torch.from_numpy(gb.call_payoff(ks)).to(dev)
                gbd, ub = gb.to(dev), torch.empty_like(ub0)
                best, sw = 1e30, 0.0
                for rep in range(3):
                    ub.copy_(ub0)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    solver.DO_timestepping(m1, m2, N, T / N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, gbd, ub, state_precision=STATE)
                    dt_ = time.perf_counter() - t1
                    if rep > 0 and dt_ < best:
                        best, sw = dt_, solver.timing()["sweep_ms"]
                solver.set_tuning("streams", 1)
                sweep_obj[str(nb) + ("_two_streams" if streams == 2 else "")] = {"value": nb * m * N / best, "sweep_only": nb * m * N / (sw * 1e-3),
                                      "sweep_frac": round(nb * m * N / (sw * 1e-3) * B_STEP / 1e9 / HBM_PEAK_GBS, 4),
                                      "kernels": solver.describe_last_sweep().split(";")[0].replace("row pass ", "")}
                del gbd, ub, ub0
            sweep_obj["ratio_512_over_256"] = round(sweep_obj["512"]["sweep_only"] / sweep_obj["256"]["sweep_only"], 4)

            # ---- single-instance latency (the literal config[1]: ONE European call) ----------------------
            g1 = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, [100.0])
            g1d, u1d = g1.to(dev), torch.from_numpy(g1.call_payoff([100.0])).to(dev)
            w1 = torch.empty_like(u1d)
            best = 1e30
            for _ in range(3):
                w1.copy_(u1d)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                solver.DO_timestepping(m1, m2, N, T / N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, g1d, w1)
                best = min(best, time.perf_counter() - t1)
            gi = H.Grid(m1, 800.0, S_0, 100.0, 20.0, m2, 5.0, V_0, 5.0 / 500)
            price1 = float(w1[0, gi.find_s_index(S_0) + gi.find_v0_index(V_0) * (m1 + 1)].item())
            single = {"wall_ms": best * 1e3, "price": price1, "kernels": solver.describe_last_sweep()}
            if (m1, m2, N) == (512, 256, 1000):
                single.update(reference_price=8.8942192888223310, price_abs_err=abs(price1 - 8.8942192888223310))

        two_streams = None
        if wl == "c3" and n_gpus == 1 and not args.skip_single and not args.tuning:
            # the same batch as two halves side by side on two streams (hadi_set_tuning "streams" = 2, an opt-in: DESIGN.md
            # section 5): one warm-up + the best of two timed passes
            solver.set_tuning("streams", 2)
            best2 = 1e30
            for rep in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                if rep > 0:
                    best2 = min(best2, time.perf_counter() - t1)
            two_streams = {"value": units_step / best2, "kernels": solver.describe_last_sweep()}
            solver.set_tuning("streams", 1)

        state_err = None
        if state == "fp32" and wl in ("c2", "c5") and n_gpus == 1 and not args.skip_single:
            # what the fp32 state costs at this step count: the K ~ 100 instance once more with the fp64 state
            gk = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, [strikes[k_mid]])
            gkd, uk = gk.to(dev), torch.from_numpy(gk.call_payoff([strikes[k_mid]])).to(dev)
            solver.DO_timestepping(m1, m2, N, T / N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, gkd, uk)
            p64 = float(uk[0, g.find_s_index(S_0) + g.find_v0_index(V_0) * (m1 + 1)].item())
            state_err = {"price_fp64_state": p64, "price_abs_diff": abs(p64 - price_check["price"]),
                         "note": "two 24-bit roundings per step accumulate like a random walk (price error 1e-7 .. 1.5e-5, erratic in N): "
                                 "the fp32 state does not guarantee a 1e-6 price tolerance"}

        # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores ------------
        if n_gpus == 1 and not args.no_cpu_baseline and wl != "c4":
            from oracle import oracle as O
            cores = O.max_threads()
            n_cpu = max(1, min(cores, 64, n_glob))
            # a bounded sample (~10-20 s wall): n_cpu instances x N_cpu of the N time steps, same step size
            budget = 6.0e7 * max(1, min(cores, n_cpu))  # point-steps at ~1e7 per core-second x ~6 s
            N_cpu = int(max(10, min(N, budget / (n_cpu * m))))
            ks = all_strikes[:n_cpu]
            gs = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, ks)
            u0c = gs.put_payoff(ks) if wl == "c3" else gs.call_payoff(ks)
            p = O.make_params(m1, m2, N_cpu, T / N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, O.AM_DIV if wl == "c3" else O.EU,
                              DIVS if wl == "c3" else None, state_fp32=1 if state == "fp32" else 0,
                              option_type=O.PUT if wl == "c3" else O.CALL, strikes=np.array(ks) if wl == "c3" else None)
            t1 = time.perf_counter()
            _, _, used = O.solve_batch(p, gs.Vec_s, gs.Vec_v, gs.Delta_s, gs.Delta_v, u0c, u0c, threads=cores)
            dt_cpu = time.perf_counter() - t1
            cpu = {"value": n_cpu * m * N_cpu / dt_cpu, "unit": "point-steps/s", "cores": int(used), "kind": "port",
                   "sample": "%d instances x %d of the %d time steps of the same %dx%d workload, OpenMP over instances, %.1f s wall"
                             % (n_cpu, N_cpu, N, m1, m2, dt_cpu)}

        desc = {
            "c2": "C2: European calls, Heston Douglas ADI, %dx%d grid, %d time steps, %d strikes per GPU (85..115), HBM-resident inputs" % (m1, m2, N, n_loc),
            "c3": "C3: American puts with 4 discrete dividends (put boundary data), %dx%d grid, %d time steps, %d strikes per GPU (85..115)" % (m1, m2, N, n_loc),
            "c5": "C5: European calls, %dx%d grid, %d time steps, %s state between the passes, %d strikes per GPU" % (m1, m2, N, state, n_loc),
            "c4": "C4: one LM iteration on a 500-option surface (50 strikes x 10 maturities, N_m = max(20, 20 T_m)), %dx%d grid: "
                  "3000-solve Jacobian + J^T J all-reduce + 500 trial solves" % (m1, m2),
        }[wl]
        out = {
            "metric": "grid-points x timesteps/sec (ADI sweep), %dx%d grid" % (m1, m2),
            "value": value, "unit": "point-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if wl == "c4" else "weak",
            "vs_baseline": None, "dtype": "f64" if state == "fp64" else "f32 state, f64 arithmetic", "data": "synthetic",
            "config": {"workload": desc, "m1": m1, "m2": m2, "timesteps": N, "instances_per_gpu": n_loc if wl != "c4" else None,
                       "instances_total": n_glob,
                       "parallelism": "instances sharded over %d GPU(s), %s" % (
                           n_gpus, "one all-reduce of 31 doubles + one of 1 double per step" if wl == "c4" else "no data-path collective"),
                       "timing_collective": collective},
            "effective_GBps": value * B_STEP / 1e9,
            "sweep_only_point_steps_per_s": (units_step / n_gpus) * args.steps / (sweep_ms * 1e-3) if wl != "c4" else None,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "batch_sweep": sweep_obj,
            "single_instance": single,
            "fp32_state_error": state_err, "two_streams": two_streams,
            "price_check": price_check,
            "device": info,
        }
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    solver.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
