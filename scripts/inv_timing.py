#!/usr/bin/env python3
"""Time of the triangular inverse inside gpbo_fit at several sizes, with a checksum of the posterior variance over 4096
candidates (the inverse factor is what the variance is computed from): python3 scripts/inv_timing.py [n ...]"""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import GP, synth
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384]
gp = GP(0)
for n in sizes:
    X = synth.uniform(1, n * 4).reshape(n, 4)
    y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(101, n)
    Xs = synth.uniform(2, 4096 * 4).reshape(4096, 4)
    best = None
    for rep in range(3):
        gp.fit(X, y, np.full(4, 0.35), 0.1)
        t = gp.timings()
        best = t['inverse_ms'] if best is None else min(best, t['inverse_ms'])
    gp.set_candidates(Xs)
    gp.posterior()
    mu, var = gp.read()
    print('n=%6d  inverse %8.3f ms  %7.2f TFLOP/s  fit %8.3f ms  var sha1 %s  var[0] %.17g' % (n, best, n ** 3 / 3 / best / 1e9, t['fit_total_ms'], hashlib.sha1(var.tobytes()).hexdigest()[:12], var[0]), flush=True)
