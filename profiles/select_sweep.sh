#!/bin/bash
# selection kernel: workgroups vs time per reference (synchronising call, 512x512, M=10000)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for g in 0 4 8 16 32 64; do
  LFD_SELECT_WORKGROUPS=$g python $REPO/bench.py --cpu-sample-refs 0 --steps 5 --spinup-s 0.02 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('workgroups %-3s select %.3f ms/ref (incl. sync)  sampled mode %.3f ms/ref' % ('$g', d['secondary_kernels']['lfd_select_filter_kernel']['ms_per_reference_incl_sync'], d['sampled_mode']['ms_per_reference']))
"
done
