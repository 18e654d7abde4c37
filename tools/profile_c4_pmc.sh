# PMC counters of BASELINE config 4 (one LM iteration on the 500-option surface: the LDS-resident small-grid kernels):
#   bash tools/profile_c4_pmc.sh [tag]      (on the GPU box, via gpurun)
# Writes profiles/<tag>_c4_pmc_summary.json and the "c4:" entry of profiles/pmc_traffic.json that bench.py --workload c4 reads for
# its VALU-issue bound (an HBM roofline does not apply to kernels whose state never leaves LDS inside the time loop).
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=gpurun_out/${TAG}_c4; mkdir -p $RAW
ARGS="bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --skip-single"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$RAW/sq1" -- python3 $ARGS > "$RAW/sq1.log" 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --output-format csv -d "$RAW/sq2" -- python3 $ARGS > "$RAW/sq2.log" 2>&1 || echo "sq2 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/kt" -- python3 $ARGS > "$RAW/bench_kt.json" 2> "$RAW/kt.log" || echo "kt failed"
python3 - "$RAW" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
raw, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(raw + "/sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hadi_small" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "bench.py --workload c4, 3 LM iterations (1 warm-up + 2 timed): per small-grid kernel the counters summed over its dispatches "
                "and divided by the number of iterations; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles"}
iters = 3
valu = 0.0
for k, d in acc.items():
    out[k] = {c: sum(v) / iters for c, v in d.items()}
    out[k]["_dispatches_per_iteration"] = len(next(iter(d.values()))) / iters
    valu += out[k].get("SQ_INSTS_VALU", 0.0)
out["valu_wave_instructions_per_iteration"] = valu
here = "profiles"
json.dump(out, open(os.path.join(here, tag + "_c4_pmc_summary.json"), "w"), indent=1, sort_keys=True)
p = os.path.join(here, "pmc_traffic.json")
rec = json.load(open(p)) if os.path.exists(p) else {}
rec["c4:50x25x500"] = {"valu_wave_instructions_per_iteration": valu,
                       "source": "profiles/%s_c4_pmc_summary.json (rocprofv3 --pmc SQ_INSTS_VALU, all hadi_small_* dispatches of one LM iteration)" % tag}
json.dump(rec, open(p, "w"), indent=1)
ks = glob.glob(os.path.join(raw, "kt", "*", "*kernel_stats.csv"))
if ks:
    import shutil
    shutil.copy(ks[0], os.path.join(here, tag + "_c4_kernel_stats.csv"))
print(json.dumps(out, indent=1)[:1500])
PY
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_c4_* profiles/pmc_traffic.json gpurun_out/profiles_out/
