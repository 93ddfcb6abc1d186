#!/bin/bash
# Inputs of the predicted 1 -> 8 GPU curve of BASELINE config 4 (56 references x 8 neighbours, `fast`): what ONE rank of an N-rank
# strong-scaling run computes - 56 / N references - measured on the one GPU of a gpurun box.  profiles/scaling_model.py adds the
# exchange from SURVEY 8e's link model.  Usage (GPU box): bash profiles/run_scaling_inputs.sh > gpurun_out/scaling_inputs.jsonl
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for n in 1 2 4 8; do
  refs=$(( (56 + n - 1) / n ))
  python $REPO/bench.py --workload config4 --scaling strong --refs $refs --light --steps 100 --cpu-sample-refs 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'ranks': $n, 'refs_per_rank': $refs, 'mode': 'dense', 'step_ms': d['ms_per_step'], 'kernel_ms': d['kernel_ms'], 'fresh_batch_ms': d['fresh_batch_ms'],
                          'points': d['survivor_fraction'] * $refs * 512 * 512}))
"
done
python $REPO/profiles/sampled_group_time.py
