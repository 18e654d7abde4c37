# PMC (HBM traffic, instruction mix) of the config-5 kernels: bash tools/profile_c5_pmc.sh   (on the GPU box, via gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in f64 f32; do
  RAW=gpurun_out/r02c5$v; mkdir -p $RAW/pmc
  if [ $v = f64 ]; then EXTRA="--workload c5 --state fp64"; BA=16.0; else EXTRA="--workload c5"; BA=8.0; fi
  EXTRA="$EXTRA" bash profiles/collect_pmc.sh $RAW/pmc 64 20 > $RAW/pmc.log 2>&1
  python3 profiles/summarize.py $RAW r02_c5$v 64 1024 512 "c5$( [ $v = f64 ] && echo f64 ):" $BA > $RAW/summary.txt 2>&1
  tail -8 $RAW/summary.txt
done
mkdir -p gpurun_out/profiles_out && cp profiles/r02_c5f64_pmc_summary.json profiles/r02_c5f32_pmc_summary.json profiles/pmc_traffic.json gpurun_out/profiles_out/
