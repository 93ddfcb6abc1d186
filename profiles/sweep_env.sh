#!/bin/bash
# sweep one environment variable over values: profiles/sweep_env.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --cpu-sample-refs 0 --steps 60 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-28s kernel_ms %.4f  frac %.4f  surv %.5f' % ('$VAR=$v', r['kernel_ms'], r['frac'], d['survivor_fraction']))
"
done
