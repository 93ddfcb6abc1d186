#!/bin/bash
# VALU instruction counts of kernel variants (build/variants/*.so), one rocprofv3 --pmc pass each
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/variants_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $REPO/build/variants/*.so; do
  name=$(basename $so .so)
  export LFD_DENSIFY_LIB=$so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$name -o p -- python3 $REPO/bench.py --steps 4 --warmup 1 --cpu-sample-refs 0 > $OUT/$name.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/$name/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("lfd_dense"):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-22s" % "$name", " ".join(f"{k[3:]}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +200k -delete
