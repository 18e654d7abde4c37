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
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=180))
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


VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4.0   # MI355X: 256 CUs x 4 SIMDs, one fp64 wave-instruction per 4 cycles at 2.4 GHz


def valu_issue_bound(seconds_per_iteration, n_gpus=1):
    """The bound of the LDS-resident small-grid kernels (config 4): fp64 VALU issue.  Wave-instructions of one LM iteration
    (rocprofv3 --pmc SQ_INSTS_VALU over all hadi_small_* dispatches, committed in profiles/pmc_traffic.json by
    tools/profile_c4_pmc.sh -- a lookup, not a measurement of this run) over the iteration's WALL time, against the chip's
    issue rate of one wave-instruction per SIMD every 4 cycles."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["c4:50x25x500"]
    except Exception:  # noqa: BLE001
        return None
    n = rec["valu_wave_instructions_per_iteration"]
    ach = n / seconds_per_iteration / n_gpus
    return {"bound": "fp64 VALU issue", "valu_wave_instructions_per_iteration": n, "achieved": ach, "peak": VALU_PEAK_WAVE_INSTS,
            "unit": "wave-instructions/s per GPU", "frac": round(ach / VALU_PEAK_WAVE_INSTS, 4), "source": rec.get("source"),
            "note": "wall time of the whole iteration (both launches, setup, host glue); the kernels' own share is higher"}


def surface_points(H):
    """50 strikes x 10 maturities, N_m = max(20, 20 T_m) (heston_calibration.cpp:2485-2531)."""
    mats = [1.0 + i * 0.25 if i < 8 else 3.0 + (i - 8) * 0.5 for i in range(10)]
    return H.make_calibration_points([S_0 * 0.75 + 1.0 * i for i in range(50)], mats)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: this process becomes the launcher.  It starts N copies of
    itself -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on
    127.0.0.1 -- relays rank 0's JSON line and exits non-zero if any rank failed (the same contract as `python -m
    torch.distributed.run --nproc-per-node N bench.py --gpus N`).  The launcher itself never imports torch and never maps
    libhadi.so (a HIP fat binary): it compiles the artefacts (hipcc / gcc need no GPU) and checks the exports with `nm -D`
    against include/hadi.h.  All ranks are supervised: the first rank that exits non-zero ends its siblings (a rank left
    alone in the rendezvous would sit there until the backend's time-out, holding the GPU lease), and the whole run has an
    upper bound (--spawn-timeout)."""
    import socket
    import subprocess
    import threading
    import __graft_entry__ as G
    G.build_libhadi()   # once, here: the ranks then find the artefacts up to date
    G.build_oracle()
    G.check_exports()
    with socket.socket() as sock:
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.spawn_timeout
    codes = [None] * len(procs)
    why = ""
    while any(c is None for c in codes):
        for r, q in enumerate(procs):
            if codes[r] is None:
                codes[r] = q.poll()
        failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if failed or time.monotonic() > deadline:
            why = ("rank %d exited with code %s first" % (failed[0], codes[failed[0]])) if failed else \
                  ("no result after %d s (--spawn-timeout)" % args.spawn_timeout)
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            t_kill = time.monotonic() + 10.0
            for r, q in enumerate(procs):
                try:
                    codes[r] = q.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    q.kill()
                    codes[r] = q.wait()
            if not failed:
                codes = [c if c else 124 for c in codes]
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    if out0 and out0[0]:
        sys.stdout.write(out0[0])
        sys.stdout.flush()
    if any(codes):
        print("bench.py: rank exit codes %s%s" % (codes, " (%s; the other ranks were ended)" % why if why else ""), file=sys.stderr)
        sys.exit(next((c for c in codes if c and c > 0), 1))
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
    ap.add_argument("--spawn-timeout", type=int, default=1500, help="--gpus N without a launcher: upper bound of the whole run in seconds")
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
    solver = H.HestonADI(dev_index)
    for kv in args.tuning:
        solver.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))

    class SweepCase:
        """One batch of a streaming workload (c2 / c3 / c5) resident in HBM: grids, initial condition, the output array."""

        def __init__(self, wl, m1, m2, N, strikes, state):
            self.wl, self.m1, self.m2, self.N, self.strikes, self.state = wl, m1, m2, N, strikes, state
            self.n, self.m = len(strikes), (m1 + 1) * (m2 + 1)
            self.BA, self.BB = (8.0, 8.0) if state == "fp32" else (16.0, 24.0) if wl == "c3" else (16.0, 16.0)
            gh = H.GridViewsBatch.for_strikes(m1, m2, S_0, V_0, strikes)
            u0 = gh.put_payoff(strikes) if wl == "c3" else gh.call_payoff(strikes)
            self.grids, self.U0 = gh.to(dev), torch.from_numpy(u0).to(dev)
            self.U = torch.empty_like(self.U0)
            self.kw = dict(state_precision=H.STATE_FP32 if state == "fp32" else H.STATE_FP64)
            if wl == "c3":
                self.kw.update(variant=H.AM_DIV, U_0=self.U0, dividends=H.Dividends(*DIVS), option_type=H.PUT, strikes=strikes)
            self.units = float(self.n) * self.m * N

        def step(self):
            self.U.copy_(self.U0)  # workspace.U <- U_0 before every call (heston_calibration.cpp:216); D2D, part of the pass.  No
            # synchronize: the launcher orders the library's stream after torch's (hadi_wait_stream)
            solver.DO_timestepping(self.m1, self.m2, self.N, T / self.N, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, self.grids, self.U, **self.kw)

        def price(self, k):
            K = self.strikes[k]
            g = H.Grid(self.m1, 8 * K, S_0, K, K / 5, self.m2, 5.0, V_0, 5.0 / 500)
            return float(self.U[k, g.find_s_index(S_0) + g.find_v0_index(V_0) * (self.m1 + 1)].item())

        def roofline(self, loop_ms):
            """One extra PROFILED pass (per-launch HIP events on the library's stream) splits `loop_ms` -- the time-loop
            events of the timed passes, per time step -- between the two kernels; the dominant kernel is the LONGER one."""
            solver.set_profiling(True)
            self.step()
            tm = solver.timing()
            path = solver.describe_last_sweep()
            solver.set_profiling(False)
            pts = float(self.n) * self.m
            a_prof = tm["pass_a_ms"] / max(1, tm["pass_a_launches"])
            b_prof = tm["pass_b_ms"] / max(1, tm["pass_b_launches"])
            share_a = tm["pass_a_ms"] / max(1e-30, tm["pass_a_ms"] + tm["pass_b_ms"])
            a_ms, b_ms = loop_ms * share_a, loop_ms * (1.0 - share_a)
            parts = path.split(";")
            names = {"a": parts[0].replace("row pass ", "").strip(), "b": (parts[1] if len(parts) > 1 else "").replace("column pass ", "").strip()}

            def leg(which, by, ms, prof):
                ach = by * pts / (ms * 1e-3) / 1e9
                return {"kernel": names[which], "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 5),
                        "avg_launch_ms_profiled": round(prof, 5), "bytes_per_launch_algorithmic": by * pts, "bytes_per_point_algorithmic": by}
            pa, pb = leg("a", self.BA, a_ms, a_prof), leg("b", self.BB, b_ms, b_prof)
            dom, dom_key = (pa, "pass_a") if a_ms >= b_ms else (pb, "pass_b")
            ach_step = (self.BA + self.BB) * pts / (loop_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom["kernel"], "dominant": dom_key, "kernels": path,
                    "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": None,
                    "avg_launch_ms": dom["avg_launch_ms"], "avg_launch_ms_profiled": dom["avg_launch_ms_profiled"],
                    "bytes_per_launch_algorithmic": dom["bytes_per_launch_algorithmic"],
                    "bytes_per_point_algorithmic": dom["bytes_per_point_algorithmic"],
                    "pass_a": pa, "pass_b": pb,
                    "sweep": {"achieved": round(ach_step, 1), "frac": round(ach_step / HBM_PEAK_GBS, 4),
                              "bytes_per_point_step": self.BA + self.BB, "loop_ms_per_time_step": round(loop_ms, 5),
                              "sweep_ms_profiled": round(tm["sweep_ms"], 3), "setup_ms": round(tm["setup_ms"], 3),
                              "finish_ms": round(tm["finish_ms"], 3)}}
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):  # HBM bytes per launch from the committed PMC run of this workload (not of this run)
                try:
                    rec = json.load(open(pmc))
                    key = "%s%s:%dx%dx%d" % (self.wl, "f64" if (self.wl == "c5" and self.state == "fp64") else "", self.m1, self.m2, self.n)
                    key = key if key in rec else "%dx%dx%d" % (self.m1, self.m2, self.n) if self.wl == "c2" else key
                    if key in rec:
                        roof["traffic"] = rec[key]["%s_bytes_per_launch" % dom_key]
                        roof["traffic_source"] = rec[key].get("source", "profiles/pmc_traffic.json")
                except Exception:  # noqa: BLE001
                    pass
            return roof

    def lm_iteration_case(pts_all, lo, hi):
        """BASELINE config 4: one LM iteration on this rank's share [lo, hi) of the 500-option surface (50x25 grids)."""
        cm1, cm2 = (m1, m2) if wl == "c4" else (50, 25)
        cm = (cm1 + 1) * (cm2 + 1)
        pts = pts_all[lo:hi]
        ks = [p.strike for p in pts]
        gh = H.GridViewsBatch.for_strikes(cm1, cm2, S_0, V_0, ks)
        grids, U0 = gh.to(dev), torch.from_numpy(gh.call_payoff(ks)).to(dev)
        market = torch.tensor([H.market.call_price(S_0, p.strike, R_D, 0.2, p.maturity) for p in pts], dtype=torch.float64, device=dev)
        ws = H.DOWorkspace(len(pts), cm, device=dev)
        holder = {"price": float("nan")}

        def step():
            J, base = solver.compute_jacobian_multi_maturity(S_0, V_0, R_D, R_F, RHO, SIGMA, KAPPA, ETA, cm1, cm2, cm, THETA,
                                                             pts, len(pts), grids, U0)
            part = comm.allreduce_sum(H.lm_partials_device(solver, J, base, market))   # RCCL, inside the timed region
            delta = H.lm_solve(part, 0.01)
            new = H.clamp_parameters(KAPPA + delta[0], ETA + delta[1], SIGMA + delta[2], RHO + delta[3], V_0 + delta[4])
            ws.U.copy_(U0)
            trial = solver.compute_base_prices_multi_maturity(S_0, new[4], R_D, R_F, new[3], new[2], new[0], new[1], cm1, cm2, cm,
                                                              THETA, pts, len(pts), grids, ws)
            solver.wait_stream()
            err = comm.allreduce_sum(np.array([float(((market - trial) ** 2).sum().item())]))
            holder["price"], holder["err"] = float(base[0].item()), float(err[0])
        units = 7.0 * cm * sum(p.time_steps for p in pts_all)   # 6 Jacobian solves + 1 trial solve per option
        return step, units, holder, ks

    # ---- this rank's shard of the workload ---------------------------------------------------------------------------
    m = (m1 + 1) * (m2 + 1)
    case = None
    if wl == "c4":
        pts_all = surface_points(H)
        lo, hi = H.shard_range(len(pts_all), n_gpus, rank, costs=[float(p.time_steps) for p in pts_all])
        step, units_step, state_holder, strikes = lm_iteration_case(pts_all, lo, hi)
        n_glob = len(pts_all)
        B_STEP = 32.0
    else:
        n_glob = n_loc * n_gpus
        all_strikes = [100.0] if n_glob == 1 else [85.0 + 30.0 * k / (n_glob - 1) for k in range(n_glob)]
        strikes = all_strikes[rank * n_loc:(rank + 1) * n_loc]
        case = SweepCase(wl, m1, m2, N, strikes, state)
        step, units_step = case.step, float(n_glob) * m * N
        B_STEP = case.BA + case.BB

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

    def timed_pass(fn, reps=2):
        """one warm-up, then the best wall time of `reps` passes (synchronised on both sides), with that pass's sweep_ms"""
        fn()
        best, sw = 1e30, 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t1
            if dt_ < best:
                best, sw = dt_, solver.timing()["sweep_ms"]
        return best, sw

    out = None
    if rank == 0:
        info = solver.device_info()
        roofline, single, sweep_obj, price_check, cpu, others, harness = None, None, None, None, None, None, None
        extras = n_gpus == 1 and not args.skip_single
        if wl != "c4":
            # price sanity inside the bench: the instance nearest K = 100
            k_mid = min(range(n_loc), key=lambda k: abs(strikes[k] - 100.0))
            price_check = {"strike": strikes[k_mid], "price": case.price(k_mid)}
            # ---- roofline of the dominant kernel, one extra profiled pass outside the timed region -----
            roofline = case.roofline(sweep_ms / max(1, args.steps) / N)  # time-loop events of the timed steps (incl. the rare dividend launches)
            roofline["sweep"]["sweep_ms"] = round(sweep_ms / max(1, args.steps), 3)
        else:
            path = solver.describe_last_sweep()
            roofline = {"bound": "hbm", "kernel": path, "achieved": round(value * 32.0 / 1e9 / n_gpus, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(value * 32.0 / 1e9 / n_gpus / HBM_PEAK_GBS, 4), "traffic": None,
                        "note": "LDS-resident kernel: the whole time loop of an instance runs out of LDS, HBM sees the state twice "
                                "per SOLVE; `achieved` is the EFFECTIVE rate at 32 B per point-step, not HBM traffic -- the bound that "
                                "applies is `valu_issue`"}
            roofline["valu_issue"] = valu_issue_bound(elapsed / args.steps, n_gpus)
            price_check = {"strike": strikes[0], "price": state_holder["price"], "trial_error": state_holder.get("err")}

        if wl == "c2" and extras:
            # ---- the same workload at the batch sizes of SURVEY.md 8(d) C2 (one warm-up + the best of two timed passes each) -----
            sweep_obj = {}
            for nb in (64, 160, 192, 256, 512):
                ks = [85.0 + 30.0 * k / (nb - 1) for k in range(nb)]
                cb = SweepCase("c2", m1, m2, N, ks, state)
                best, sw = timed_pass(cb.step)
                sweep_obj[str(nb)] = {"value": nb * m * N / best, "sweep_only": nb * m * N / (sw * 1e-3),
                                      "sweep_frac": round(nb * m * N / (sw * 1e-3) * B_STEP / 1e9 / HBM_PEAK_GBS, 4),
                                      "kernels": solver.describe_last_sweep()}
                del cb
            sweep_obj["ratio_512_over_256"] = round(sweep_obj["512"]["sweep_only"] / sweep_obj["256"]["sweep_only"], 4)

            # ---- single-instance latency (the literal config[1]: ONE European call) ----------------------
            c1 = SweepCase("c2", m1, m2, N, [100.0], "fp64")
            best, _ = timed_pass(c1.step, reps=3)
            price1 = c1.price(0)
            single = {"wall_ms": best * 1e3, "price": price1, "kernels": solver.describe_last_sweep()}
            if (m1, m2, N) == (512, 256, 1000):
                single.update(reference_price=8.8942192888223310, price_abs_err=abs(price1 - 8.8942192888223310))
            del c1

        if wl == "c2" and extras and not args.tuning and not (args.m1 or args.m2 or args.timesteps or args.instances):
            # ---- the other BASELINE configs on the same line (outside the timed region): one warm-up + the best of two
            # timed passes + one profiled pass each; `value` = whole job incl. setup / pack / unpack, wall clock ----------
            others = {}
            for key, (owl, ostate) in (("c3", ("c3", "fp64")), ("c5_fp64", ("c5", "fp64")), ("c5_fp32", ("c5", "fp32"))):
                om1, om2, oN, on, _, _ = WORKLOADS[owl]
                oks = [85.0 + 30.0 * k / (on - 1) for k in range(on)]
                oc = SweepCase(owl, om1, om2, oN, oks, ostate)
                best, sw = timed_pass(oc.step)
                k100 = min(range(on), key=lambda k: abs(oks[k] - 100.0))
                price = oc.price(k100)
                roof = oc.roofline(sw / oN)
                others[key] = {"workload": "%dx%d grid, %d time steps, %d instances, %s%s state" % (
                                   om1, om2, oN, on, "American puts with 4 discrete dividends, " if owl == "c3" else "European calls, ", ostate),
                               "value": oc.units / best, "unit": "point-steps/s", "ms_per_batch": round(best * 1e3, 3),
                               "sweep_only_point_steps_per_s": oc.units / (sw * 1e-3),
                               "pass_a": roof["pass_a"], "pass_b": roof["pass_b"], "sweep": roof["sweep"], "dominant": roof["dominant"],
                               "kernels": roof["kernels"], "price_K%g" % oks[k100]: price}
                del oc
            if "c5_fp32" in others and "c5_fp64" in others:
                p32 = [v for k, v in others["c5_fp32"].items() if k.startswith("price_K")][0]
                p64 = [v for k, v in others["c5_fp64"].items() if k.startswith("price_K")][0]
                others["c5_fp32"]["price_abs_diff_vs_fp64_state"] = abs(p32 - p64)
            pts4 = surface_points(H)
            step4, units4, hold4, _ = lm_iteration_case(pts4, 0, len(pts4))
            best4, _ = timed_pass(step4, reps=5)
            others["c4"] = {"workload": "one LM iteration on the 500-option surface (50 strikes x 10 maturities, 50x25 grids): 3000-solve "
                                        "Jacobian + device-side normal equations + 500 trial solves",
                            "ms_per_iteration": round(best4 * 1e3, 3), "value": units4 / best4, "unit": "point-steps/s",
                            "kernels": solver.describe_last_sweep(), "trial_error": hold4.get("err"), "valu_issue": valu_issue_bound(best4),
                            "note": "LDS-resident kernels: no HBM roofline applies (the state crosses HBM twice per SOLVE)"}

            # ---- the reference's own perf harness (src/perfomance_test.cpp:20-230): 50x25 grid, N = 20, every strike 85,
            # instances {1 .. 500}, mean of 10 runs after one warm-up; and its American + dividend twin (README.md:16) ------
            harness = {"shape": "m1=50, m2=25, N=20, all strikes 85 (src/perfomance_test.cpp:46-57,141-210), mean of 10 runs",
                       "reference_published": {"single_european_s": 0.003, "american_dividend_500_s": 0.02, "hardware": "NVIDIA A100",
                                               "source": "README.md:14-18 (grid and step count not stated; context only)"},
                       "european": {}, "american_dividend": {}}
            hm1, hm2, hN = 50, 25, 20
            for hv, hkey in ((H.EU, "european"), (H.AM_DIV, "american_dividend")):
                for ni in ((1, 10, 20, 50, 100, 200, 300, 500) if hv == H.EU else (500,)):
                    gh = H.GridViewsBatch.for_strikes(hm1, hm2, S_0, V_0, [85.0] * ni)
                    u0 = torch.from_numpy(gh.call_payoff([85.0] * ni)).to(dev)
                    gd, uu = gh.to(dev), torch.empty_like(u0)
                    kw = dict(variant=H.AM_DIV, U_0=u0, dividends=H.Dividends(*DIVS)) if hv == H.AM_DIV else {}

                    def hstep():
                        uu.copy_(u0)
                        solver.DO_timestepping(hm1, hm2, hN, T / hN, THETA, R_D, R_F, RHO, SIGMA, KAPPA, ETA, gd, uu, **kw)
                    hstep()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    ksum = 0.0
                    for _ in range(10):
                        hstep()
                        ksum += solver.timing()["sweep_ms"]
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t1) / 10
                    harness[hkey][str(ni)] = {"wall_ms": round(wall * 1e3, 4), "kernel_ms": round(ksum / 10, 4),
                                              "point_steps_per_s": ni * 51 * 26 * hN / wall}

        two_streams = None
        state_err = None
        if state == "fp32" and wl in ("c2", "c5") and extras:
            # what the fp32 state costs at this step count: the K ~ 100 instance once more with the fp64 state
            ck = SweepCase(wl, m1, m2, N, [strikes[k_mid]], "fp64")
            ck.step()
            p64 = ck.price(0)
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
            "other_workloads": others,
            "reference_harness": harness,
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
