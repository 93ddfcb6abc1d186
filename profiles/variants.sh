#!/bin/bash
# A/B timing of kernel variants: profiles/variants.sh build "<name>:<flags>" ...   (here, no GPU needed)
#                                profiles/variants.sh run                          (on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$REPO/lichtfeld-densification-plugin_amd/csrc
VD=$REPO/build/variants
mkdir -p $VD
if [ "$1" = "build" ]; then
  shift
  [ "${LFD_VARIANTS_KEEP:-0}" = "1" ] || rm -f $VD/*.so
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    [ "$flags" = "$spec" ] && flags=""
    sched="-mllvm -amdgpu-sched-strategy=max-ilp"
    case "$flags" in *amdgpu-sched-strategy*) sched="";; esac      # a variant may bring its own scheduler strategy
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-fast-math -Wno-unused-function $sched $flags \
        -pthread $SRC/lfd_api.hip $SRC/lfd_kernels.hip $SRC/lfd_select.hip $SRC/lfd_writer.hip $SRC/lfd_host.hip $SRC/lfd_image.hip -o $VD/$name.so || echo "BUILD FAILED $name" ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
  done
  wait
  ls -la $VD
else
  shift
  # (round 3: `run --light --steps 400` - the bench's own sustained loop without the heavy side legs - repeats to +-0.3 %; the default
  # of 60 steps after a full pass carries the launch-time transient and reads ~10 % higher, comparable only within one log)
  for so in $VD/*.so; do
    name=$(basename $so .so)
    LFD_DENSIFY_LIB=$so python $REPO/bench.py --cpu-sample-refs 0 --steps 60 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('[lfd]'): print('  ' + l.strip())
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-24s kernel_ms %.4f  frac %.4f  surv %.5f' % ('$name', r['kernel_ms'], r['frac'], d['survivor_fraction']))
"
  done
fi
