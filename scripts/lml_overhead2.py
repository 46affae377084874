import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_optimization_amd import GP, synth
gp = GP(0)
for n in (3000, 4096, 8192):
    X = synth.uniform(7, n * 3).reshape(n, 3)
    y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2
    for B in (16, 64):
        th = np.full((B, 3), 0.5) + 0.001 * np.arange(B)[:, None]
        sg = np.full(B, 0.05)
        gp.lml_batch(X, y, th, sg, grad=True)
        t0 = time.perf_counter()
        gp.lml_batch(X, y, th, sg, grad=True)
        dt = time.perf_counter() - t0
        print('n=%5d B=%4d: %.2f ms per call, %.3f ms per setting' % (n, B, dt * 1e3, dt * 1e3 / B), flush=True)
