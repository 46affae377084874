#!/bin/bash
# A/B of the lattice loop (cfg4) between builds of the library on one box: bash scripts/ab_cfg4.sh <variant.so>...
# prints ms per iteration and the first picks of every build, two rounds
for rep in 1 2; do
  for lib in "" "$@"; do
    GPBO_LIBRARY=$lib timeout -k 10 200 python bench.py --config cfg4 --iters 200 --warmup 1 2>/dev/null > gpurun_out/ab4.json || exit 1
    python3 -c "
import json,sys,os; r=json.load(open('gpurun_out/ab4.json')); print(os.path.basename(sys.argv[1]) or 'current', round(r['ms_per_step'],4), r['roofline'].get('avg_launch_ms'), r['config'].get('picked_head'))" "$lib"
  done
done
