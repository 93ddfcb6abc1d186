#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [bench args]
# Writes rocprofv3 summaries under gpurun_out/prof_<tag>/ : kernel-trace stats, then PMC passes
# (SQ counters, FETCH_SIZE and WRITE_SIZE in separate passes as the TCC slots require).
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --light $*"
# the kernel-trace pass times what bench.py times: many back-to-back launches, short spin-up
TRACE_ARGS="--steps 400 --warmup 5 --spinup-s 0.02 --light $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $TRACE_ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem -o mem -- python3 $REPO/bench.py $ARGS > $OUT/pmc_mem.log 2>&1
python3 $REPO/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep what is judged (stats + our kernels' rows), drop the bulky per-dispatch traces of torch's kernels
for f in $(find $OUT -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  head -1 $f > $f.tmp; grep "lfd_" $f >> $f.tmp; mv $f.tmp $f
done
find $OUT -name "*.db" -delete
du -sh $OUT
