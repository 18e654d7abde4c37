#!/usr/bin/env python3
"""Turns raw rocprofv3 output (gpurun_out/..., scratch) into the committed summaries under profiles/.

    python profiles/summarize.py <raw_dir> <tag> [instances] [m1] [m2] [key prefix] [algorithmic bytes per point and pass]

raw_dir holds kt/ (rocprofv3 --kernel-trace --stats of bench.py) and pmc/{sq1,sq2,fetch,write}
(profiles/collect_pmc.sh).  Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json and
updates profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM-traffic convention (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream, so read bytes =
2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def per_kernel(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        d[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_dispatches": len(next(iter(cs.values())))}
            for k, cs in d.items()}


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    inst = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    m1 = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    m2 = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    prefix = sys.argv[6] if len(sys.argv) > 6 else ""   # key prefix in pmc_traffic.json, e.g. "c5f64:" (bench.py looks it up)
    bytes_alg = float(sys.argv[7]) if len(sys.argv) > 7 else 16.0  # algorithmic bytes per point and pass
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(raw, "kt", "*", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(here, tag + "_kernel_stats.csv"))
    bj = os.path.join(raw, "bench_kt.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(here, tag + "_bench_under_rocprof.json"))
    summary = {"_note": "mean per dispatch over the profiled run; SQ_* in the counters' native units "
                        "(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles)",
               "workload": "%dx%d grid, %d instances, bench.py --timesteps 20" % (m1, m2, inst)}
    for sub in ("sq1", "sq2", "fetch", "write"):
        f = glob.glob(os.path.join(raw, "pmc", sub, "*", "*counter_collection.csv"))
        if not f:
            continue
        for k, v in per_kernel(f[0]).items():
            if "hadi_pass" in k:
                summary.setdefault(k, {}).update(v)
    pts = inst * (m1 + 1) * (m2 + 1)
    traffic = {}
    for k, v in summary.items():
        if not isinstance(v, dict) or "FETCH_SIZE" not in v:
            continue
        rd = 2.0 * v["FETCH_SIZE"] * 1024.0
        wr = v.get("WRITE_SIZE", 0.0) * 1024.0
        v["hbm_read_bytes_corrected"] = rd
        v["hbm_write_bytes"] = wr
        v["hbm_bytes_per_point"] = (rd + wr) / pts
        v["algorithmic_bytes_per_point"] = bytes_alg
        if "TCC_HIT_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        traffic[k] = rd + wr
    json.dump(summary, open(os.path.join(here, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    # (a workload may run more than one kernel per pass -- the American sweeps take the explicit pair on the first and the
    # dividend steps: the kernel with the most dispatches is the one the bench line's roofline is about)
    disp = lambda k: summary[k].get("_dispatches", 0)
    pa = [traffic[k] for k in sorted((k for k in traffic if "pass_a" in k), key=disp, reverse=True)]
    pb = [traffic[k] for k in sorted((k for k in traffic if "pass_b" in k), key=disp, reverse=True)]
    if pa:
        path = os.path.join(here, "pmc_traffic.json")
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec["%s%dx%dx%d" % (prefix, m1, m2, inst)] = {
            "pass_a_bytes_per_launch": pa[0], "pass_b_bytes_per_launch": pb[0] if pb else None,
            "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE x2 on gfx950)" % tag}
        json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (round(v["hbm_bytes_per_point"], 2) if isinstance(v, dict) and "hbm_bytes_per_point" in v else None)
                      for k, v in summary.items()}, indent=1))


if __name__ == "__main__":
    main()
