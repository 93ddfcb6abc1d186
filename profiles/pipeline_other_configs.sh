#!/bin/bash
# The pipeline leg (densify.dense_init end to end, bench_pipeline.py) on the shapes of the other BASELINE configurations - on the GPU box from the repo root:
#   bash profiles/pipeline_other_configs.sh > gpurun_out/pipeline_other_configs.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() {   # name, args...
  name=$1; shift
  timeout 400 python3 $REPO/bench_pipeline.py --latency-ms 0 "$@" 2>>$REPO/gpurun_out/pipeline_other_configs.err | python3 -c "
import json, sys
p = json.load(sys.stdin)['pipeline']
sc = p['scene']
print('== $name: %d cameras %dx%d, grid %dx%d, %d references x %d neighbours = %d pairs' % (sc['cameras'], sc['image_size'][0], sc['image_size'][1], sc['grid'][0], sc['grid'][1], sc['references'], sc['neighbours'], sc['pairs']))
for mode in ('sampled', 'dense'):
    for k, v in p[mode].items():
        if 'seconds' in v:
            print('   %-8s %-34s %7.3f s  %7.1f refs/s  %8.1f pairs/s  %10.4g points/s  (%d points, file %.1f MB)' % (mode, k, v['seconds'], v['refs_per_s'], v['pairs_per_s'], v['points_per_s'], v['points'], v['file_bytes'] / 1e6))
    st = p[mode].get('stages')
    if st: print('   %-8s ms per reference by stage: %s' % (mode, st['ms_per_reference']))
print('   pcie', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.get('pcie', {}).items() if k != 'what'})
"
}
run "config 2 (garden @fast, GUI defaults)" --cams 185 --size 1297x840 --setting fast
run "config 2 CLI defaults (0.75 of the cameras x 4 neighbours)" --cams 185 --size 1297x840 --setting fast --num-refs 0.75 --nns 4
run "config 3 (bicycle @high: 960^2 grid over 640-px match images)" --cams 194 --size 1237x822 --setting high --refs-per-launch 8
run "config 4 (garden, ref-fraction 0.3, 8 neighbours)" --cams 185 --size 1297x840 --setting fast --num-refs 0.3 --nns 8
run "config 5 (precise 1280^2, ROI of 12 cameras, 8 neighbours)" --cams 12 --size 1297x840 --setting precise --num-refs 0.8 --nns 8 --refs-per-launch 4
