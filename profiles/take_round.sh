#!/bin/bash
# One box, one lease: everything the round's numbers come from (run on the GPU box from the repo root: bash profiles/take_round.sh r5).
#   1. rocprofv3 --kernel-trace --stats of the bench's timed loop, the PMC passes (SQ, FETCH_SIZE, WRITE_SIZE in separate passes), the instruction mix
#   2. profiles/traffic.json regenerated from them (what `roofline.traffic` / `roofline.valu` of the bench line quote)
#   3. the full default `python bench.py` (headline, side legs, cpu_baseline, parity, the pipeline leg) -> gpurun_out/<tag>/bench.json
# The summaries land in gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>/ .
set -u
TAG=${1:-r5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
bash $REPO/profiles/run_profile.sh $TAG > $OUT/run_profile.log 2>&1
bash $REPO/profiles/run_mix.sh ${TAG}_mix > $OUT/run_mix.log 2>&1
cp $REPO/gpurun_out/prof_$TAG/summary.txt $OUT/summary.txt
cp $REPO/gpurun_out/prof_${TAG}_mix/summary.txt $OUT/instruction_mix.txt
f=$(find $REPO/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
python3 $REPO/profiles/make_traffic_json.py $REPO/gpurun_out/prof_$TAG $REPO/gpurun_out/prof_${TAG}_mix profiles/$TAG > $OUT/traffic.json
cp $OUT/traffic.json $REPO/profiles/traffic.json
cd $REPO && python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
r = d["roofline"]
print("value %.4g pts/s  step %.4f ms  kernel %.4f ms  frac %.4f  traffic/algo %s  valu frac %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("traffic_over_algorithmic"), (r.get("valu") or {}).get("frac")))
print("default mode", d.get("default_mode"))
p = d.get("pipeline", {})
for mode in ("sampled", "dense"):
    for k, v in p.get(mode, {}).items():
        if isinstance(v, dict) and "seconds" in v:
            print(mode, k, "%.3f s  %.1f refs/s  %.1f pairs/s  %.4g pts/s" % (v["seconds"], v["refs_per_s"], v["pairs_per_s"], v["points_per_s"]))
print("pcie", p.get("pcie"))
PY
grep lfd_dense_kernel $OUT/kernel_stats.csv | head -2
