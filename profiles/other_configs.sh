#!/bin/bash
# dense kernel on the other BASELINE.json configurations (synthetic stand-ins, same generator as bench.py)
# (round 4: steady state - 400 steps after the spin-up, the dense kernel's own start / stop events - not the 30-step runs of round 3, which sat inside the clock transient)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  python $REPO/bench.py --light --steps ${LFD_OTHER_STEPS:-400} --cpu-sample-refs 0 --parity-refs 0 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-44s kernel_ms %.4f  cells/s %.3e  points/s %.3e  pairs/s %.3e  alg GB/s %.0f  frac %.3f  surv %.4f' % ('$*', r['kernel_ms'], d['cells_per_s'], d['value'], d['pairs_per_s'], r['achieved'], r['frac'], d['survivor_fraction']))
"
}
run --preset turbo --k 1 --refs 64
run --preset fast --k 3 --refs 64
run --preset fast --k 4 --refs 64
run --preset fast --k 8 --refs 56
run --preset high --k 3 --refs 32
run --preset precise --k 8 --refs 12
# the BASELINE configurations by name (round 6: config3 = bicycle's 194 cameras 1237x822 @high added as a workload)
run --workload config2
run --workload config3
run --workload config4
run --workload config5
