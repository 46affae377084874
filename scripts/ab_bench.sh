#!/bin/bash
# A/B timing of several builds of the library on the same box: bash scripts/ab_bench.sh <variant.so>... (two rounds)
for rep in 1 2; do
  for lib in "" "$@"; do
    GPBO_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/ab.json || exit 1
    python3 -c "
import json,sys,os; r=json.load(open('gpurun_out/ab.json')); print(os.path.basename(sys.argv[1]) or 'baseline', r['value'], r['roofline']['achieved'])" "$lib"
  done
done
