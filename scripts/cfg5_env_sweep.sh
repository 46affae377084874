#!/bin/bash
# scripts/cfg5_env_sweep.sh VAR "v1 v2 ...": the 32-setting likelihood sweep once per value of an environment switch
V=$1; VALS=$2
for t in $VALS; do
  env $V=$t timeout -k 10 200 python3 bench.py --config cfg5 --settings 32 --steps 1 --warmup 1 2>/dev/null > gpurun_out/ab.json || exit 1
  python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], sys.argv[2], r['value'])" "$V" "$t"
done
