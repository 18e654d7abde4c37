#!/bin/bash
# Collects PMC counters for the bench workload in separate rocprofv3 passes (counters only with
# --kernel-trace, as the pool requires).  Run on the GPU box from the repo root:
#   bash profiles/collect_pmc.sh <outdir> [instances] [timesteps]
set -u
OUT=${1:-gpurun_out/pmc}; INST=${2:-256}; TS=${3:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# EXTRA: further bench.py arguments (e.g. EXTRA="--workload c5 --state fp64")
ARGS="bench.py --steps 1 --warmup 0 --instances $INST --timesteps $TS --no-cpu-baseline --skip-single ${EXTRA:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
  --output-format csv -d "$OUT/sq1" -- python3 $ARGS > "$OUT/sq1.log" 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/sq2" -- python3 $ARGS > "$OUT/sq2.log" 2>&1 || echo "sq2 failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1 || echo "fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1 || echo "write failed"
find "$OUT" -name "*counter_collection.csv" | head
