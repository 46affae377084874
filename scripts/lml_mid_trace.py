#!/usr/bin/env python3
"""One likelihood sweep (-logL + gradient) of B settings at a mid size, for a kernel trace: python3 scripts/lml_mid_trace.py [n] [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import GP, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = synth.uniform(7, n * 3).reshape(n, 3)
y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, 2] + 0.05 * synth.normal(8, n)
th = np.full((B, 3), 0.5) + 0.003 * np.arange(B)[:, None]
sg = np.full(B, 0.05)
gp = GP(0)
gp.lml_batch(X, y, th, sg, grad=True)
t0 = time.perf_counter()
for _ in range(5):
    gp.lml_batch(X, y, th, sg, grad=True)
print('n=%d B=%d: %.3f ms per sweep' % (n, B, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
