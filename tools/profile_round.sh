set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=gpurun_out/r01b; mkdir -p $RAW/kt $RAW/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-single > $RAW/bench_kt.json 2> $RAW/kt.log || echo "kt failed"
bash profiles/collect_pmc.sh $RAW/pmc 256 20 > $RAW/pmc.log 2>&1
python3 profiles/summarize.py $RAW r01 256 512 256 > $RAW/summary.txt 2>&1
cat $RAW/summary.txt | tail -20
python3 bench.py > $RAW/bench_full.json 2> $RAW/bench_full.err; tail -1 $RAW/bench_full.json | cut -c1-1500
cp $RAW/bench_full.json profiles/r01_bench.json
python3 tools/bench_configs.py > profiles/r01_configs.json 2>$RAW/cfg.err
mkdir -p gpurun_out/profiles_out && cp profiles/r01_* profiles/pmc_traffic.json gpurun_out/profiles_out/
