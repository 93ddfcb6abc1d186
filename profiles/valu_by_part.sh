#!/bin/bash
# vector instructions per launch of the dense kernel with parts compiled out (variants built by profiles/variants.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for so in $REPO/build/variants/*.so; do
  name=$(basename $so .so)
  export LFD_DENSIFY_LIB=$so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $REPO/gpurun_out/vp_$name -o p -- python3 $REPO/bench.py --steps 4 --warmup 1 --spinup-s 0.02 --cpu-sample-refs 0 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$REPO/gpurun_out/vp_$name/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("lfd_dense"):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-18s" % "$name", "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
  rm -rf $REPO/gpurun_out/vp_$name
done
