#!/bin/bash
# The lattice loop (cfg4) with the new row of W and alpha written by the sweep launch's first workgroup (default) or by a
# launch of their own in front of the sweep (GPBO_APPEND_IN_SWEEP=0), same binary, two rounds
for rep in 1 2; do
  for v in 1 0; do
    GPBO_APPEND_IN_SWEEP=$v timeout -k 10 200 python bench.py --config cfg4 --iters 200 --warmup 1 2>/dev/null > gpurun_out/ab4.json || exit 1
    python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab4.json')); print('in_sweep', sys.argv[1], round(r['ms_per_step'],4), r['roofline'].get('avg_launch_ms'), r['config'].get('picked_head')[:4])" $v
  done
done
