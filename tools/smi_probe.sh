#!/bin/bash
# run the bench for a while and sample clocks / power
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --steps 40 --warmup 1 --no-cpu-baseline --skip-single > gpurun_out/bw.json 2>gpurun_out/bw.err &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ';'; echo
  sleep 1.5
done
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/bw.json').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['roofline']['avg_launch_ms'], d['roofline']['pass_b']['avg_launch_ms'])"
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ';'; echo
