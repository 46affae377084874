import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_optimization_amd import GP, synth
gp = GP(0)
n, B = 500, 32
X = synth.uniform(7, n * 3).reshape(n, 3)
y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2
th = np.full((B, 3), 0.5) + 0.01 * np.arange(B)[:, None]
sg = np.full(B, 0.05)
for _ in range(3):
    gp.lml_batch(X, y, th, sg, grad=True)
