# Kernel durations of the single-instance run (512x256x1000, one European call): bash tools/profile_single.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=gpurun_out/single; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -- python3 tools/single_instance.py > $RAW/log.txt 2>&1 || echo failed
tail -4 $RAW/log.txt
f=$(find $RAW/kt -name "*kernel_stats.csv" | head -1); head -6 $f
