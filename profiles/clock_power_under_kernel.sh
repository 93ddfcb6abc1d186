#!/bin/bash
# What the chip's clock and power do while the dense kernel runs back to back (bash profiles/clock_power_under_kernel.sh, on the GPU box): rocm-smi polled beside a
# sustained run of the bench's timed loop.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $REPO/bench.py --light --cpu-sample-refs 0 --parity-refs 0 --steps 30000 --warmup 10 > /tmp/cp_bench.json 2>/tmp/cp_bench.err &
BP=$!
sleep 4
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
python3 -c "
import json
d=json.load(open('/tmp/cp_bench.json')); r=d['roofline']; print('sustained run: kernel_ms', round(r['kernel_ms'],4), 'steps', d['steps'], 'ms_per_step', round(d['ms_per_step'],4))"
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | tr -s ' '
