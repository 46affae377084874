#!/bin/bash
# Tile-order sweep of the candidate kernel on one box: column-group width (column blocks of 128 candidates per group)
# against the chunk (candidates per launch).
for c in ${CFGS:-cfg2}; do
  for rep in 1 2; do
    for ch in ${CHUNKS:-32768}; do
      for cg in ${GROUPS_:-100000 64 32 16 8}; do
        GPBO_TRMM_COLGROUP=$cg timeout -k 10 300 python bench.py --config $c --chunk $ch --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/cg.json || exit 1
        python3 -c "
import json,sys; r=json.load(open('gpurun_out/cg.json')); print(sys.argv[1], 'chunk', sys.argv[3], 'colgroup', sys.argv[2], r['value'], r['roofline']['achieved'])" $c $cg $ch
      done
    done
  done
done
