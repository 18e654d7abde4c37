#!/usr/bin/env python3
"""Diagnostic: run bench.py against an alternative build of libhadi (tools/_var_<name>.so)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pde_based_heston_solver_gpu_accelerated_amd._native as nat
nat.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
