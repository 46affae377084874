#!/bin/bash
# candidates per variance launch vs throughput at the 32768-point configuration (131072-candidate slice)
for c in 4096 8192 16384 32768; do
  timeout -k 10 300 python bench.py --config cfg3 --candidates 131072 --steps 2 --warmup 1 --no-cpu-baseline --chunk $c 2>/dev/null > gpurun_out/chunk3_$c.json || exit 1
  python3 -c "
import json,sys; r=json.load(open(sys.argv[1])); print(sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['achieved'], r['phases_ms_per_step'])" gpurun_out/chunk3_$c.json $c
done
