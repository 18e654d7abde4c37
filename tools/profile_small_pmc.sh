# PMC counters of the LDS-resident small-grid kernel: bash tools/profile_small_pmc.sh <outdir> <instances>   (on the GPU box)
set -u
OUT=${1:-gpurun_out/pmc_small}; INST=${2:-1536}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
export SIZES=$INST STEPS=100
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$OUT/sq1" -- python3 tools/small_latency.py > "$OUT/sq1.log" 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/sq2" -- python3 tools/small_latency.py > "$OUT/sq2.log" 2>&1 || echo "sq2 failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "small" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
