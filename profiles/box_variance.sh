#!/bin/bash
# one line per box: sustained clock under the dense kernel's instruction mix (profiles/microbench/clock_under_load) beside the dense kernel's time
# (bench.py --light): bash profiles/box_variance.sh >> gpurun_out/box_variance.txt     (one gpurun call = one box of the pool)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
clk=$(./profiles/microbench/clock_under_load | awk '/dense kernel.s mix/ {for (i = 1; i <= NF; ++i) if ($i == "p50") print $(i + 1)}')
f64=$(./profiles/microbench/clock_under_load | awk '/^v_fma_f64, 8 waves/ {for (i = 1; i <= NF; ++i) if ($i == "p50") print $(i + 1)}')
python bench.py --light --cpu-sample-refs 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('box %s  clock p50: mix %s MHz, f64 %s MHz | kernel_ms mean %.4f p50 %.4f min %.4f | ms_per_step %.4f | mean x mix clock = %.1f' % (
            '$(hostname)', '$clk', '$f64', r['kernel_ms'], r['kernel_ms_percentiles']['p50'], r['kernel_ms_percentiles']['min'], d['ms_per_step'], r['kernel_ms'] * float('$clk')))
"
