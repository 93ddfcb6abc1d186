#!/bin/bash
# memory-pipeline counters of the dense kernel (TA / TCP / TCC latency, stall and busy counters): profiles/pmc_mem.sh <tag> [env assignments...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_LFIFO_FULL_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "GRBM_TC_BUSY GRBM_UTCL2_BUSY" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  t0=$(date +%s)
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $REPO/bench.py --steps 3 --warmup 1 --spinup-s 0.02 --cpu-sample-refs 0 --light > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc $? in $(( $(date +%s) - t0 )) s" >> $OUT/progress.txt
  find $OUT/p$i -name "*.db" -delete
  for f in $(find $OUT/p$i -name "*counter_collection.csv"); do head -1 $f > $f.tmp; grep "lfd_dense" $f >> $f.tmp; mv $f.tmp $f; done
  find $OUT/p$i -name "*.csv" -size +2000k -delete
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("lfd_dense"):
            acc[(row["Kernel_Name"][:28], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"{k[0]:28s} {k[1]:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +100k -delete
