#!/usr/bin/env python3
"""Randomised differential run of the 'backtrack' step rule: the descent inside one launch (gpbo_descent_small, steprule 2)
against the host loop that evaluates every trial point through lml_batch.  Same iterates, lengths and exit flags expected, bit
for bit; prints the wall time of both.  python3 scripts/fuzz_backtrack.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import notebook_api as nb

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = ran = 0
worst = 0.0
shrunk = 0
t_dev = t_host = 0.0
for c in range(cases):
    n = int(rng.choice([1, 2, 7, 31, 32, 33, 50, 64, 65, 100, 127, 128]))
    d = int(rng.choice([1, 2, 3, 5, 8, 16]))
    nruns = int(rng.choice([1, 3, 10]))
    its = int(rng.choice([1, 15, 40]))
    norm = str(rng.choice(['component', 'full', 'none']))
    alpha = float(rng.choice([0.01, 0.1, 0.5, 2.0]))
    beta = float(rng.choice([0.7, 1.0]))
    sY = float(rng.choice([0.0, 0.02, 0.2]))
    kernel = int(rng.integers(0, 2))
    X = rng.random((n, d)); y = np.sin(4 * X.sum(1)) + 0.05 * rng.standard_normal(n)
    x0 = rng.random((nruns, d)) + 0.05
    res = []
    for dev in ('1', '0'):
        os.environ['GPBO_DESCENT_DEVICE'] = dev
        t0 = time.perf_counter()
        res.append(nb.hypopt(nruns, its, alpha, beta, norm, 'backtrack', X, y.reshape(-1, 1), None, sY, x0=x0.copy(), kernel=kernel))
        if dev == '1':
            t_dev += time.perf_counter() - t0
        else:
            t_host += time.perf_counter() - t0
    ran += 1
    (wa, ca, fa), (wb, cb, fb) = res
    same = list(fa) == list(fb)
    for r in range(nruns):
        same = same and len(wa[r]) == len(wb[r]) and np.array_equal(np.array(wa[r]), np.array(wb[r]), equal_nan=True)
        if same:
            fin = np.isfinite(np.array(cb[r]))
            same = same and np.array_equal(np.isfinite(np.array(ca[r])), fin)
            if fin.any():
                worst = max(worst, float(np.max(np.abs(np.array(ca[r])[fin] - np.array(cb[r])[fin]) / np.maximum(np.abs(np.array(cb[r])[fin]), 1.0))))
            if len(wb[r]) > 2:
                s = np.linalg.norm(np.diff(np.array(wb[r]), axis=0), axis=1)
                shrunk += int(np.isfinite(s).all() and s[-1] < 0.5 * s[0])
    if not same or worst > 1e-12:
        bad += 1
        print('MISMATCH', dict(n=n, d=d, nruns=nruns, its=its, norm=norm, alpha=alpha, beta=beta, sY=sY, kernel=kernel), flush=True)
print('backtrack descent: cases %d  mismatches %d  worst relative cost difference %.2e  starts whose step shrank %d  device %.2f s  host loop %.2f s'
      % (ran, bad, worst, shrunk, t_dev, t_host), flush=True)
sys.exit(1 if bad else 0)
