#!/bin/bash
# A/B of two builds of the library on the lattice loop (cfg4 --lattice): scripts/ab_lattice.sh <other libgpbo.so> [runs]
# Prints ms per iteration of each build, alternating, and whether the picks agree with the streaming route.
set -e
other=$1; runs=${2:-3}
cd "$(dirname "$0")/.."
cp bayesian_optimization_amd/libgpbo.so /tmp/libgpbo_head.so
for r in $(seq 1 $runs); do
  for which in head other; do
    if [ $which = head ]; then cp /tmp/libgpbo_head.so bayesian_optimization_amd/libgpbo.so; else cp "$other" bayesian_optimization_amd/libgpbo.so; fi
    timeout -k 10 200 python3 bench.py --config cfg4 --lattice --no-cpu-baseline --steps 1 --warmup 1 > /tmp/ab_line.json
    python3 -c "import json;d=json.load(open('/tmp/ab_line.json'));print('$which', d['ms_per_step'], d['value'], d['roofline']['frac'])"
  done
done
cp /tmp/libgpbo_head.so bayesian_optimization_amd/libgpbo.so
