#!/bin/bash
# bench.py's sharded leg on ONE rank through RCCL (no link: the exchange's own software cost), config 4; usage: run_forced_dist.sh [bench args ...]
export LFD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
python bench.py --gpus 1 --workload config4 --steps 50 --light "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['replication']
        print('args', ' '.join(sys.argv[1:]), '| replicated', r['n_replicated'], '| step ms', round(d['ms_per_step'],4), 'pure', round(d['pure_sharding_ms'],4), 'compute', round(d['compute_ms'],4),
              'exposed', round(d['exchange_ms_exposed'],4), '| resident ms', round(d['exchange']['points']/d['value_sharded_resident']*1e3,4), '| rounds', d['exchange']['rounds'])
" "$@"
