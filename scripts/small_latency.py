"""Latency of one notebook-sized BO step (n = 94, m = 6080): fit, posterior, EI + argmax; GPU vs the NumPy oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_optimization_amd import GP, notebook_api as nb
from oracle import gp_oracle as o

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g1_mat_mixture.npz'))
X, y, Xs, th = g['xTrain'], g['yTrain'], g['xTest'], g['theta']
gp = GP(0)
for rep in range(3):
    t0 = time.perf_counter(); gp.fit(X, y, th, 0.0); t1 = time.perf_counter()
    gp.set_candidates(Xs); t2 = time.perf_counter()
    gp.posterior(); t3 = time.perf_counter()
    gp.acquire(0, 0.01); t4 = time.perf_counter()
print('GPU object API: fit %.2f ms, set_candidates %.2f, posterior %.2f, acquire %.2f ms' % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
for rep in range(3):
    t0 = time.perf_counter()
    Mu, Cov, *_ = nb.gprfcn(Xs, X, y, th, 0.0)
    ei = nb.expected_improvement(Mu, Cov, 0.01)
    jj = int(np.argmax(ei))
    t1 = time.perf_counter()
print('GPU notebook API (gprfcn + expected_improvement + argmax): %.2f ms' % (1e3 * (t1 - t0)))
t0 = time.perf_counter()
mu, var, _ = o.gprfcn_diag(Xs, X, y, th, 0.0)
ei = o.expected_improvement(mu, var, 0.01)
t1 = time.perf_counter()
print('NumPy oracle (diag only): %.2f ms' % (1e3 * (t1 - t0)))
