#!/usr/bin/env python3
"""One-rank rehearsal of bench.py's process-group set-up on a GPU box: gloo default group + RCCL sub-group with a probe
all-reduce, barrier and Communicator traffic on it (the N > 1 case needs one GPU per rank)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch
import bench
import pde_based_heston_solver_gpu_accelerated_amd as H
torch.cuda.set_device(0)
args = types.SimpleNamespace(backend="nccl", allow_gloo=False)
dist, group, name = bench.init_distributed(args, torch, 0)
print("collective:", name, "group backend:", dist.get_backend(group))
comm = H.Communicator(group=group)
print("communicator device:", comm.device, "backend:", comm.backend)
print("allreduce:", comm.allreduce_sum(np.arange(31.0))[:4], "gather:", comm.allgather_rows(np.ones((3, 5)), [3]).shape)
dist.barrier(group=group)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group); print("max:", t.item())
dist.destroy_process_group()
print("nccl probe ok")
