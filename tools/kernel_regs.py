#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in libhadi, from the compiler's own metadata (hipcc --save-temps).
    python tools/kernel_regs.py [filter-substring ...]
Build-time check, no GPU needed."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc", "hadi_api.hip")

def collect(extra_flags=()):
    tmp = tempfile.mkdtemp(prefix="hadi_regs_")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "--save-temps", *extra_flags,
                           "-o", os.path.join(tmp, "x.so"), SRC], cwd=tmp, stderr=subprocess.DEVNULL)
    s = open(os.path.join(tmp, "hadi_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = []
    for b in s.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(HadiSweepArgs, int)", "")
        out.append((dn, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    return out, s

if __name__ == "__main__":
    rows, _ = collect()
    for dn, v, sg, sp, scr, lds in sorted(rows):
        if len(sys.argv) > 1 and not any(f in dn for f in sys.argv[1:]):
            continue
        print("%-78s vgpr %3d sgpr %3d spill %3d scratch %4d B" % (dn[:78], v, sg, sp, scr))
