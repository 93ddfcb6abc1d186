export LFD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for rep in auto 20; do
python bench.py --gpus 1 --workload config4 --steps 50 --light --replicate $rep 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps(d['replication'])); print('value',d['value'],'pure',d['value_pure_sharding'],'compute_only',d['value_compute_only'],'resident',d['value_sharded_resident'],'ms',d['ms_per_step'],d['pure_sharding_ms'],d['compute_ms'], 'exposed', d['exchange_ms_exposed'], d['exchange']['allgather_GBps_per_peer_measured'], d['exchange']['collective_ms_measured'])
"
done
