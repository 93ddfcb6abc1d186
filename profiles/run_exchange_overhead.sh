#!/bin/bash
# host-side profile of one step of bench.py's sharded leg on ONE rank through RCCL (no link: what is left is the software)
export LFD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LFD_BENCH_CPROFILE=1
python bench.py --gpus 1 --workload config4 --steps 50 --light --replicate 0 2>&1 >/dev/null | grep -v "^\[W\|amdgpu.ids" | cut -c1-200
