#!/usr/bin/env python3
"""Cholesky time and rate of gpbo_fit at several sizes (RBF, d = 4, sigmaY = 0.1): python3 scripts/chol_timing.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import GP, synth
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]
gp = GP(0)
for n in sizes:
    X = synth.uniform(1, n * 4).reshape(n, 4)
    y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(101, n)
    best = None
    for rep in range(3):
        gp.fit(X, y, np.full(4, 0.35), 0.1)
        t = gp.timings()
        c = t['chol_ms'] / t['chol_count']
        best = c if best is None else min(best, c)
    print('n=%6d  cholesky %8.3f ms  %7.2f TFLOP/s   inverse %7.3f ms  trsv %6.3f ms  fit %8.3f ms  theta1 %.12g nll %.12g'
          % (n, best, n ** 3 / 3 / best / 1e9, t['inverse_ms'], t['trsv_ms'], t['fit_total_ms'], gp.theta1, gp.nll), flush=True)
