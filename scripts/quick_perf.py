import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from bayesian_optimization_amd import GP, synth
cfg = dict(synth.CONFIGS['cfg2'])
m = int(os.environ.get('M', 1 << 19))
gp = GP(0)
print('selftest', gp.selftest_gemm(4096, rate=True), flush=True)
X, y, theta = synth.training_set(cfg)
gp.fit(X, y, theta, cfg['sigmaY'])
gp.set_candidates(synth.candidates(cfg, 0, m))
for chunk in [int(c) for c in os.environ.get('CHUNKS', '0').split(',')]:
    gp.set_chunk(chunk)
    gp.posterior()
    gp.posterior()
    tm = gp.timings()
    print('order', os.environ.get('GPBO_TRMM_ORDER', '1'), 'chunk', chunk, 'trmm TF %.2f' % (tm['trmm_flops'] / tm['trmm_ms'] / 1e9), 'posterior ms %.2f ks %.2f trmm %.2f launches %d' % (tm['posterior_total_ms'], tm['ksbuild_ms'], tm['trmm_ms'], tm['trmm_launches']), flush=True)
