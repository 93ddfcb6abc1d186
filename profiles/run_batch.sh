#!/bin/bash
# dense kernel against the number of reference views per launch (fast 512^2, k = 3): bash profiles/run_batch.sh   (on the GPU box)
for r in 8 16 32 64 128 256; do
  python bench.py --light --steps 30 --refs $r 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('refs per launch %4d   kernel_ms %.4f  us per reference %.3f  points/s %.3e  frac %.4f' % ($r, r['kernel_ms'], r['kernel_ms'] * 1e3 / $r, d['value'], r['frac']))
"
done
