import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from bayesian_optimization_amd import notebook_api as nb
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g1_mat_mixture.npz'))
xT, yT = g['xTrain'], g['yTrain']
np.random.seed(0)
for nruns, its in ((10, 100), (200, 50)):
    t0 = time.time(); th = nb.hypopt_wrap(nruns, its, 0.01, 1, 'component', 'normal', xT, yT, None, 0.0); t1 = time.time()
    print('hypopt_wrap nruns=%d max_its=%d: %.2fs (%.2f ms per -logL+grad) theta=%s nll=%.4f' % (nruns, its, t1 - t0, (t1 - t0) / (nruns * (its + 1)) * 1e3, th, nb.mlefcn(xT, yT, None, th, 0.0)), flush=True)
# mid-size training sets go through the general multi-launch path (several lanes)
from bayesian_optimization_amd import synth
for n in (200, 500, 1000):
    X = synth.uniform(7, n * 3).reshape(n, 3)
    y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, 2] + 0.05 * synth.normal(8, n)
    np.random.seed(1)
    t0 = time.time(); th = nb.hypopt_wrap(32, 30, 0.01, 1, 'component', 'normal', X, y, None, 0.05); t1 = time.time()
    print('n=%d: hypopt_wrap nruns=32 max_its=30: %.2fs (%.2f ms per -logL+grad) theta=%s' % (n, t1 - t0, (t1 - t0) / (32 * 31) * 1e3, th), flush=True)
