#!/bin/bash
# candidates per variance launch vs throughput on the default workload
for c in 16384 32768 65536 131072; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunk $c 2>/dev/null > gpurun_out/chunk_$c.json || exit 1
  python3 -c "
import json,sys; r=json.load(open(sys.argv[1])); print(sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['achieved'])" gpurun_out/chunk_$c.json $c
done
