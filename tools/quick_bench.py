#!/usr/bin/env python3
"""Kernel-level A/B numbers on the GPU box: python tools/quick_bench.py [c2 c3 c5 c5f64 am512 ...]
Prints per workload the row-pass / column-pass mean launch time (HIP events, profiled step) and the whole-sweep rate."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "c2": ["--workload", "c2"], "c2_64": ["--workload", "c2", "--instances", "64"], "c2_160": ["--workload", "c2", "--instances", "160"],
    "c2_512": ["--workload", "c2", "--instances", "512"],
    "c3": ["--workload", "c3"], "c3strip": ["--workload", "c3", "--tuning", "strip=1"],
    "g256": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "1024"],
    "g256s": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "1024", "--tuning", "strip=1"],
    "g256_300": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "300"],
    "g256_300s": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "300", "--tuning", "strip=1"],
    "g200": ["--workload", "c2", "--m1", "200", "--m2", "100", "--timesteps", "400", "--instances", "700"],
    "g200s": ["--workload", "c2", "--m1", "200", "--m2", "100", "--timesteps", "400", "--instances", "700", "--tuning", "strip=1"],
    "c2_64s": ["--workload", "c2", "--instances", "64", "--tuning", "strip=1"], "c2_160s": ["--workload", "c2", "--instances", "160", "--tuning", "strip=1"],
    "c2_128": ["--workload", "c2", "--instances", "128"], "c2_128s": ["--workload", "c2", "--instances", "128", "--tuning", "strip=1"],
    "c2_320": ["--workload", "c2", "--instances", "320"], "c2_384": ["--workload", "c2", "--instances", "384"],
    "c2_600": ["--workload", "c2", "--instances", "600"], "c2_768": ["--workload", "c2", "--instances", "768"],
    "c5f64_g6": ["--workload", "c5", "--state", "fp64", "--tuning", "col_groups=6"], "c5f64_g9": ["--workload", "c5", "--state", "fp64", "--tuning", "col_groups=9"],
    "c5f64_g17": ["--workload", "c5", "--state", "fp64", "--tuning", "col_groups=17"], "c5f64_g3": ["--workload", "c5", "--state", "fp64", "--tuning", "col_groups=3"],
    "c5f64_r24": ["--workload", "c5", "--state", "fp64", "--tuning", "row_tile=24"], "c5f64_r64": ["--workload", "c5", "--state", "fp64", "--tuning", "row_tile=64"],
    "c5f64_r32": ["--workload", "c5", "--state", "fp64", "--tuning", "row_tile=32"],
    **{"c5f64_%d%s" % (k, t): ["--workload", "c5", "--state", "fp64", "--instances", str(k), "--timesteps", "400", "--tuning", "strip=%d" % v]
       for k in (8, 16, 32, 48, 64, 96, 128, 192, 256) for t, v in (("s", 1), ("r", 0))},
    **{"c2_64g%d" % g: ["--workload", "c2", "--instances", "64", "--tuning", "col_groups=%d" % g] for g in (2, 3, 4, 5, 8, 9)},
    **{"c2_128g%d" % g: ["--workload", "c2", "--instances", "128", "--tuning", "col_groups=%d" % g] for g in (2, 3, 4, 5, 8, 9)},
    "c2_g1": ["--workload", "c2", "--tuning", "col_groups=1"], "c2_g2": ["--workload", "c2", "--tuning", "col_groups=2"], "c2_g9": ["--workload", "c2", "--tuning", "col_groups=9"],
    **{"c2_%d_sb%d" % (k, b): ["--workload", "c2", "--instances", str(k), "--tuning", "strip=1", "--tuning", "strip_blocks=%d" % b]
       for k in (32, 64, 96, 128, 160, 192) for b in (1, 2, 3, 4, 5, 6)},
    **{"c3_sb%d" % b: ["--workload", "c3", "--tuning", "strip_blocks=%d" % b] for b in (1, 2, 3, 4)},
    # American puts with dividends on the 1024x512 grid, 64 instances: paired strips / shared ring, P representation / explicit pair
    **{"w_am%s%s" % (t, u): ["--workload", "c3", "--m1", "1024", "--m2", "512", "--timesteps", "300", "--instances", "64",
                            "--tuning", "strip=%d" % v, "--tuning", "american_p=%d" % w]
       for t, v in (("s", 1), ("r", 0)) for u, w in (("", 1), ("x", 0))},
    **{"c2_%d_t2" % k: ["--workload", "c2", "--instances", str(k), "--tuning", "streams=2"] for k in (32, 64, 96, 128, 160, 192, 256, 320, 384, 512)},
    **{"c2_%d_t1" % k: ["--workload", "c2", "--instances", str(k)] for k in (32, 64, 96, 128, 160, 192, 256, 320, 384, 512)},
    "c3_t2": ["--workload", "c3", "--tuning", "streams=2"], "c5f64_t2": ["--workload", "c5", "--state", "fp64", "--tuning", "streams=2"], "c5_t2": ["--workload", "c5", "--tuning", "streams=2"],
    **{"%s_t%d" % (nm, t): base + (["--tuning", "streams=2"] if t == 2 else [])
       for t in (1, 2) for nm, base in {
           "g256": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "1024"],
           "g256_300": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "300"],
           "g200": ["--workload", "c2", "--m1", "200", "--m2", "100", "--timesteps", "400", "--instances", "700"],
           "g128": ["--workload", "c2", "--m1", "128", "--m2", "64", "--timesteps", "400", "--instances", "2000"],
           "g100": ["--workload", "c2", "--m1", "100", "--m2", "50", "--timesteps", "400", "--instances", "3000"],
           "c2_16": ["--workload", "c2", "--instances", "16"], "c2_48": ["--workload", "c2", "--instances", "48"],
           "c2_24": ["--workload", "c2", "--instances", "24"], "c2_96": ["--workload", "c2", "--instances", "96"],
           "c2_192": ["--workload", "c2", "--instances", "192"], "c2_320": ["--workload", "c2", "--instances", "320"],
           "c3_256": ["--workload", "c3", "--instances", "256"], "c3_1024": ["--workload", "c3", "--instances", "1024"],
       }.items()},
    **{"c2_%d_t2s%d" % (k, b): ["--workload", "c2", "--instances", str(k), "--tuning", "streams=2", "--tuning", "strip=1"] + (["--tuning", "strip_blocks=%d" % b] if b else [])
       for k in (32, 48, 64, 96, 128) for b in (0, 1, 2, 3, 4, 6)},
    "c2_t2b1": ["--workload", "c2", "--tuning", "streams=2", "--tuning", "strip_blocks=1"],
    "c2_t2b1g3": ["--workload", "c2", "--tuning", "streams=2", "--tuning", "strip_blocks=1", "--tuning", "col_groups=3"],
    "c2_t2b1g2": ["--workload", "c2", "--tuning", "streams=2", "--tuning", "strip_blocks=1", "--tuning", "col_groups=2"],
    "c2_512_t2b1g3": ["--workload", "c2", "--instances", "512", "--tuning", "streams=2", "--tuning", "strip_blocks=1", "--tuning", "col_groups=3"],
    **{"%s_p0" % nm: base + ["--tuning", "pair_strips=0"] for nm, base in {
           "c3": ["--workload", "c3"], "c3_256": ["--workload", "c3", "--instances", "256"], "c3_1024": ["--workload", "c3", "--instances", "1024"],
           "g256": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "1024"],
           "g256_300": ["--workload", "c2", "--m1", "256", "--m2", "128", "--timesteps", "500", "--instances", "300"],
           "g200": ["--workload", "c2", "--m1", "200", "--m2", "100", "--timesteps", "400", "--instances", "700"]}.items()},
    "c3_256": ["--workload", "c3", "--instances", "256"], "c3_1024": ["--workload", "c3", "--instances", "1024"],
    "c3_t2p1": ["--workload", "c3", "--tuning", "streams=2", "--tuning", "pair_strips=1"],
    "c3_1024_t2": ["--workload", "c3", "--instances", "1024", "--tuning", "streams=2"],
    "c5f64_pf0": ["--workload", "c5", "--state", "fp64", "--tuning", "col_prefetch=0"], "c5_pf0": ["--workload", "c5", "--tuning", "col_prefetch=0"],
    "c5f64_128": ["--workload", "c5", "--state", "fp64", "--instances", "128", "--timesteps", "500"],
    "c5f64_128_pf0": ["--workload", "c5", "--state", "fp64", "--instances", "128", "--timesteps", "500", "--tuning", "col_prefetch=0"],
    "g700x300": ["--workload", "c2", "--m1", "700", "--m2", "300", "--timesteps", "500", "--instances", "128"],
    "g700x300_pf0": ["--workload", "c2", "--m1", "700", "--m2", "300", "--timesteps", "500", "--instances", "128", "--tuning", "col_prefetch=0"],
    **{"%s_il0" % nm: base + ["--tuning", "tile_interleave=0"] for nm, base in {
        "c2": ["--workload", "c2"], "c3": ["--workload", "c3"], "c5": ["--workload", "c5"], "c5f64": ["--workload", "c5", "--state", "fp64"],
        "c2_64": ["--workload", "c2", "--instances", "64"], "c2_128": ["--workload", "c2", "--instances", "128"],
        "c2am": ["--workload", "c3", "--m1", "512", "--m2", "256", "--timesteps", "1000", "--instances", "256"],
        "g700x300": ["--workload", "c2", "--m1", "700", "--m2", "300", "--timesteps", "500", "--instances", "128"],
        "c5f64_pf0": ["--workload", "c5", "--state", "fp64", "--tuning", "col_prefetch=0"]}.items()},
    **{"%s_gr" % nm: base + ["--tuning", "graph_max_melems=4096"] for nm, base in {
        "c2": ["--workload", "c2"], "c3": ["--workload", "c3"], "c5f64": ["--workload", "c5", "--state", "fp64"], "c2_64": ["--workload", "c2", "--instances", "64"],
        "c2_128": ["--workload", "c2", "--instances", "128"]}.items()},
    "c2ring": ["--workload", "c2", "--tuning", "strip=0"], "c2am": ["--workload", "c3", "--m1", "512", "--m2", "256", "--timesteps", "1000", "--instances", "256"], "c5": ["--workload", "c5"], "c5f64": ["--workload", "c5", "--state", "fp64"], "c4": ["--workload", "c4"],
}
argv = sys.argv[1:]
libs = [None]
if "--lib" in argv:  # --lib a.so,b.so : every case is timed with each library in turn (same box, interleaved)
    k = argv.index("--lib"); libs = argv[k + 1].split(","); del argv[k:k + 2]
reps = 1
if "--reps" in argv:
    k = argv.index("--reps"); reps = int(argv[k + 1]); del argv[k:k + 2]
for name, lib in [(n, l) for n in (argv or ["c2"]) for _ in range(reps) for l in libs]:
    head = [sys.executable, os.path.join(ROOT, "bench.py")] if lib in (None, "head") else \
           [sys.executable, os.path.join(ROOT, "tools", "bench_variant.py"), os.path.join(ROOT, lib)]
    out = subprocess.run(head + ["--steps", "2", "--warmup", "1", "--skip-single", "--no-cpu-baseline"] + CASES[name],
                         capture_output=True, text=True)
    name = name if lib in (None,) else "%s[%s]" % (name, os.path.basename(lib).replace("_var_", "").replace(".so", ""))
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", out.stderr[-800:]); continue
    d = json.loads(line[-1]); r = d["roofline"]
    pa = r.get("pass_a") or {}; pb = r.get("pass_b") or {}; sw = r.get("sweep") or {}
    print("%-16s value %.4e  row %.5f ms (%.3f)  col %.5f ms (%.3f)  sweep %.3f | %s" % (
        name, d["value"], pa.get("avg_launch_ms") or 0, pa.get("frac") or 0, pb.get("avg_launch_ms") or 0, pb.get("frac") or 0, sw.get("frac") or 0,
        (r.get("kernels") or r["kernel"])[:100]), flush=True)
