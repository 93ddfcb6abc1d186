#!/bin/bash
# dense kernel at 7 ... 2 resident workgroups per CU (dynamic LDS padding): bash profiles/run_occ.sh   (on the GPU box)
for lds in 0 4000 10000 18000 32000 58000; do
  LFD_DENSE_EXTRA_LDS=$lds python bench.py --light --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
n = 163840 // (22472 + $lds)
print('extra LDS %6d -> %d workgroups per CU   kernel_ms %.4f  frac %.4f' % ($lds, min(n, 7), r['kernel_ms'], r['frac']))
"
done
