# PMC (HBM traffic, instruction mix) of the config-3 and config-5 kernels: bash tools/profile_c5_pmc.sh [c3|c5|all] [tag]   (on the GPU box, via gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, bench arguments, instances, m1, m2, key prefix, algorithmic bytes per point and pass
  RAW=gpurun_out/$1; mkdir -p $RAW/pmc
  EXTRA="$2" bash profiles/collect_pmc.sh $RAW/pmc $3 20 > $RAW/pmc.log 2>&1
  python3 profiles/summarize.py $RAW $1 $3 $4 $5 "$6" $7 > $RAW/summary.txt 2>&1
  tail -6 $RAW/summary.txt
}
TAG=${2:-r03}
case "${1:-all}" in
  c3|all) run ${TAG}_c3 "--workload c3" 512 256 128 "c3:" 16.0 ;;&
  c5|all) run ${TAG}_c5f64 "--workload c5 --state fp64" 64 1024 512 "c5f64:" 16.0
          run ${TAG}_c5f32 "--workload c5" 64 1024 512 "c5:" 8.0 ;;
esac
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_c*_pmc_summary.json profiles/pmc_traffic.json gpurun_out/profiles_out/
