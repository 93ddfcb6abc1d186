#!/bin/bash
# all-warps variants (build/variants/aw*.so) on the headline shape, turbo k=1 and fast k=2 / k=4
for so in build/variants/*.so; do
  name=$(basename $so .so)
  for cfg in "--preset turbo --k 1 --refs 64" "--preset fast --k 2 --refs 64" "--preset fast --k 3 --refs 64" "--preset fast --k 4 --refs 64"; do
    LFD_DENSIFY_LIB=$so python bench.py --light --steps 40 $cfg 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-8s %-34s kernel_ms %.4f  frac %.3f' % ('$name', '$cfg', r['kernel_ms'], r['frac']))
"
  done
done
