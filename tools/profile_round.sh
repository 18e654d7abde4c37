# Regenerates the committed profile summaries on the GPU box: bash tools/profile_round.sh <tag>   (e.g. r02)
# Raw rocprofv3 output goes to gpurun_out/<tag>/ (scratch); the summaries land in profiles/ and are copied to
# gpurun_out/profiles_out/ so that they come back with the gpurun merge.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=gpurun_out/$TAG; mkdir -p $RAW/kt $RAW/pmc $RAW/kt_c3 $RAW/kt_c5 $RAW/kt_c5f64
# kernel trace + stats of the headline workload (same command as the bench line, fewer repetitions)
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-single > $RAW/bench_kt.json 2> $RAW/kt.log || echo "kt failed"
# PMC passes (counters only with --kernel-trace), 20 time steps
bash profiles/collect_pmc.sh $RAW/pmc 256 20 > $RAW/pmc.log 2>&1
python3 profiles/summarize.py $RAW $TAG 256 512 256 > $RAW/summary.txt 2>&1
tail -20 $RAW/summary.txt
# kernel stats of the other bench workloads (config 3, config 5 with fp32 and fp64 state)
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt_c3 -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --skip-single > $RAW/bench_c3_kt.json 2> $RAW/kt_c3.log || echo "kt c3 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt_c5 -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --skip-single > $RAW/bench_c5_kt.json 2> $RAW/kt_c5.log || echo "kt c5 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt_c5f64 -- python3 bench.py --workload c5 --state fp64 --steps 1 --warmup 1 --no-cpu-baseline --skip-single > $RAW/bench_c5f64_kt.json 2> $RAW/kt_c5f64.log || echo "kt c5f64 failed"
for w in c3 c5 c5f64; do f=$(find $RAW/kt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${TAG}_${w}_kernel_stats.csv; cp $RAW/bench_${w}_kt.json profiles/${TAG}_${w}_bench_under_rocprof.json; done
# config 4: VALU instruction count of one LM iteration (the bound of the LDS-resident kernels on the bench lines below)
bash tools/profile_c4_pmc.sh $TAG > $RAW/c4_pmc.log 2>&1
# the plain bench lines of this round (no profiler attached)
python3 bench.py > profiles/${TAG}_bench.json 2> $RAW/bench_full.err
python3 bench.py --workload c3 > profiles/${TAG}_bench_c3.json 2> $RAW/bench_c3.err
python3 bench.py --workload c5 > profiles/${TAG}_bench_c5.json 2> $RAW/bench_c5.err
python3 bench.py --workload c4 --steps 5 > profiles/${TAG}_bench_c4.json 2> $RAW/bench_c4.err
# PMC (HBM traffic, instruction mix) of the config-3 and config-5 kernels
bash tools/profile_c5_pmc.sh all $TAG
# the column pass against its alternatives and its own memory traffic; one / two / automatic streams; grids beyond the streaming kernels
python3 tools/colpass_ab.py > profiles/${TAG}_colpass_ab.txt 2> $RAW/colpass_ab.err
python3 tools/stream_sweep.py > profiles/${TAG}_stream_sweep.txt 2> $RAW/stream_sweep.err
python3 tools/seq_bench.py > profiles/${TAG}_seq_bench.txt 2> $RAW/seq_bench.err
# Craig-Sneyd on the strips / on the shared ring against Douglas (512x256 x256, 700x300 x128, 1024x512 x64, 256x128 x512)
(python3 tools/cs_bench.py 256 100; python3 tools/cs_bench.py 128 100 700 300; python3 tools/cs_bench.py 64 60 1024 512; python3 tools/cs_bench.py 512 100 256 128) 2> $RAW/cs_bench.err | grep -v amdgpu.ids > profiles/${TAG}_cs_bench.txt
# the literal config 2 (ONE 512x256x1000 instance) and small batches: instance-resident launch against the streaming path
python3 tools/team_ab.py 1 2 4 8 > profiles/${TAG}_team_ab.txt 2> $RAW/team_ab.err
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* profiles/pmc_traffic.json gpurun_out/profiles_out/
ls gpurun_out/profiles_out
