"""Cholesky / inverse time vs leading dimension: power-of-two n_pad against n_pad + 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_optimization_amd import GP, synth

gp = GP(0)
for n in [int(a) for a in sys.argv[1:]] or [4096, 4200, 8192, 8300, 16384, 16500]:
    d = 8
    X = synth.uniform(5, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1))
    for rep in range(2):
        gp.fit(X, y, np.full(d, 0.5), 0.1)
    t = gp.timings()
    c = t['chol_ms'] / max(t['chol_count'], 1)
    print('n=%6d n_pad=%6d chol %.2f ms (%.1f TF)  inverse %.2f ms (%.1f TF)  trsv %.2f  fit %.1f' % (
        n, -(-n // 128) * 128, c, n ** 3 / 3 / c / 1e9, t['inverse_ms'], n ** 3 / 3 / t['inverse_ms'] / 1e9, t['trsv_ms'], t['fit_total_ms']), flush=True)
