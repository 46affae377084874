#!/usr/bin/env python3
"""Replays the descent cases of scripts/fuzz_more_vs_oracle.py (same generator state) and prints, for every case whose
device-resident and host-loop iterates differ, where they part and by how much.  python3 scripts/descent_diff_probe.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import notebook_api as nb
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
for c in range(cases):
    n = int(rng.choice([1, 2, 7, 31, 32, 33, 50, 64, 65, 100, 127, 128]))
    d = int(rng.choice([1, 2, 3, 5, 8, 16]))
    nruns = int(rng.choice([1, 3, 10]))
    its = int(rng.choice([1, 15, 60]))
    norm = str(rng.choice(['component', 'full', 'none']))
    step = str(rng.choice(['normal', 'diminishing']))
    sY = float(rng.choice([0.0, 0.02, 0.2]))
    X = rng.random((n, d)); y = np.sin(4 * X.sum(1)) + 0.05 * rng.standard_normal(n)
    x0 = rng.random((nruns, d)) + 0.05
    beta = float(rng.choice([0.7, 1.0]))
    res = []
    try:
        for dev in ('1', '0'):
            os.environ['GPBO_DESCENT_DEVICE'] = dev
            res.append(nb.hypopt(nruns, its, 0.01, beta, norm, step, X, y.reshape(-1, 1), None, sY, x0=x0.copy()))
    except np.linalg.LinAlgError:
        continue
    (wa, ca, fa), (wb, cb, fb) = res
    for r in range(nruns):
        A, B = np.array(wa[r]), np.array(wb[r])
        if A.shape == B.shape and np.array_equal(A, B, equal_nan=True):
            continue
        k = 0
        while k < min(len(A), len(B)) and np.array_equal(A[k], B[k], equal_nan=True):
            k += 1
        print(dict(case=c, n=n, d=d, norm=norm, step=step, sY=sY, beta=beta, start=r), 'lengths', len(A), len(B), 'flags', fa[r], fb[r], 'first difference at iterate', k)
        if k < min(len(A), len(B)):
            print('   device', A[k], '\n   host  ', B[k], '\n   rel diff', np.max(np.abs(A[k] - B[k]) / np.abs(B[k])), 'costs before', ca[r][k - 1], cb[r][k - 1])
print('done')
