#!/bin/bash
# Same-box A/B of the 256 x 128 tile of the variance product (GPBO_TRMM_TALL=1) against the product's 128 x 128 tile:
# bit-equality of the posterior first, then the cfg2 line and the cfg3 block twice each way, alternating.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3_tall_tile_ab.txt
: > $O
timeout -k 10 120 python3 - >> $O 2>&1 <<'PY' || { tail -5 $O; exit 1; }
import os, subprocess, sys, numpy as np
code = '''
import sys, numpy as np
sys.path.insert(0, ".")
from bayesian_optimization_amd import GP, synth
cfg = dict(synth.CONFIGS["cfg2"], n=int(sys.argv[2]))
X, y, th = synth.training_set(cfg)
g = GP(0); g.fit(X, y, th, cfg["sigmaY"], kernel=cfg["kernel"]); g.set_candidates_uniform(7, 0, 40000); g.posterior()
mu, var = g.read(); np.save(sys.argv[1], var)
'''
for n in (512, 4096):
    out = []
    for env in ({}, {'GPBO_TRMM_TALL': '1'}):
        f = '/tmp/var_%d.npy' % len(out)
        subprocess.run([sys.executable, '-c', code, f, str(n)], env=dict(os.environ, **env), check=True)
        out.append(np.load(f))
    print('n=%d variance bit-identical: %s' % (n, np.array_equal(out[0], out[1])))
    assert np.array_equal(out[0], out[1])
PY
for rep in 1 2; do
  for tall in 0 1; do
    if [ $tall = 1 ]; then export GPBO_TRMM_TALL=1; else unset GPBO_TRMM_TALL; fi
    timeout -k 10 200 python3 bench.py --no-cfg45 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1])
print('tall=$tall rep=$rep cfg2 frac %.4f (%.3f ms/launch)  cfg3 frac %.4f (%.2f ms/launch)  argmax %d/%d' % (b['roofline']['frac'], b['roofline']['avg_launch_ms'], b['cfg3']['roofline']['frac'], b['cfg3']['roofline']['avg_launch_ms'], b['argmax']['index'], b['cfg3']['argmax']['index']))" >> $O || exit 1
  done
done
cat $O
