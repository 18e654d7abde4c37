#!/usr/bin/env python3
"""Builds an alternative libhadi for same-box A/B runs: python tools/build_variant.py <name> [-DFLAG ...] -> tools/_var_<name>.so
(git-ignored; travels to the GPU box with the snapshot).  tools/quick_bench.py --lib tools/_var_<name>.so ... times it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_var_%s.so" % name)
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", *flags, "-o", out,
                       os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc", "hadi_api.hip")])
print(out)
