#!/bin/bash
# The lattice loop (cfg4) with an environment switch of the library on (1) and off (0), same binary, two rounds:
#   bash scripts/ab_cfg4_env.sh GPBO_APPEND_IN_SWEEP   new row of W and alpha inside the sweep launch / in a launch of their own
VAR=${1:-GPBO_APPEND_IN_SWEEP}
for rep in 1 2; do
  for v in 1 0; do
    env $VAR=$v timeout -k 10 200 python bench.py --config cfg4 --iters 200 --warmup 1 2>/dev/null > gpurun_out/ab4.json || exit 1
    python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab4.json')); print(sys.argv[1], sys.argv[2], round(r['ms_per_step'],4), r['roofline'].get('avg_launch_ms'), r['config'].get('picked_head')[:4])" $VAR $v
  done
done
