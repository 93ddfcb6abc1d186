#!/bin/bash
# quick counter pass over the dense kernels: profiles/pmc_quick.sh <tag> [env assignments...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_FLAT" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $REPO/bench.py --steps 4 --warmup 1 --cpu-sample-refs 0 > $OUT/p$i.log 2>&1
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("lfd_dense"):
            acc[(row["Kernel_Name"][:28], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"{k[0]:28s} {k[1]:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +100k -delete
