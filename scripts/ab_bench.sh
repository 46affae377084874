#!/bin/bash
# A/B timing of two builds of the library on the same box: bash scripts/ab_bench.sh <variant.so> [bench args]
V=$1; shift
for rep in 1 2; do
  for lib in "" "$V"; do
    GPBO_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/ab.json || exit 1
    python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab.json')); print(sys.argv[1] or 'baseline', r['value'], r['roofline']['achieved'])" "$lib"
  done
done
