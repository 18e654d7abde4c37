#!/usr/bin/env python3
"""Single-instance wall time (512x256x1000, one European call) for one or more builds: python tools/single_ab.py a.so b.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for lib in sys.argv[1:]:
        code = ("import sys,os,time; sys.path.insert(0,%r); import pde_based_heston_solver_gpu_accelerated_amd._native as nat; nat.LIB_PATH=%r;"
                "import numpy as np, torch; import pde_based_heston_solver_gpu_accelerated_amd as H;"
                "dev=torch.device('cuda:0'); s=H.HestonADI(0); m1,m2,N=512,256,1000;"
                "g=H.GridViewsBatch.for_strikes(m1,m2,100.0,0.04,[100.0]); U0=torch.from_numpy(g.call_payoff([100.0])).to(dev); gd=g.to(dev); U=torch.empty_like(U0);"
                "best=1e9\n"
                "for _ in range(5):\n"
                "    U.copy_(U0); torch.cuda.synchronize(); t=time.perf_counter(); s.DO_timestepping(m1,m2,N,1.0/N,0.8,0.025,0.0,-0.9,0.3,1.5,0.04,gd,U); best=min(best,time.perf_counter()-t)\n"
                "print('%%-22s wall %%.2f ms  price %%.12f' %% (os.path.basename(nat.LIB_PATH), best*1e3, float(U[0,181+78*513].item())))") % (ROOT, os.path.join(ROOT, lib))
        subprocess.run([sys.executable, "-c", code])
