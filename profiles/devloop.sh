#!/bin/bash
# quick GPU check used while tuning: parity tests of the kernels, then the headline bench without the CPU leg
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
python bench.py --cpu-sample-refs 0 --steps 100 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('kernel_ms', round(r['kernel_ms'], 4), 'frac', round(r['frac'], 4), 'pts/s', '%.3e' % d['value'], 'surv', round(d['survivor_fraction'], 5))
"
