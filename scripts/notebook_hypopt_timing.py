"""The notebook's own optimiser settings (BayesOpt_v4.ipynb cells: nruns=10 / max_its=1e3 and nruns=200 / max_its=500,
alpha=0.01, beta=1, 'component', 'normal') on the 94-point mixture data."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from bayesian_optimization_amd import notebook_api as nb
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g1_mat_mixture.npz'))
xT, yT = g['xTrain'], g['yTrain']
np.random.seed(0)
nb.hypopt_wrap(2, 2, 0.01, 1, 'component', 'normal', xT, yT, None, 0.0)      # warm-up
for nruns, its in ((10, 1e3), (200, 500)):
    t0 = time.time(); th = nb.hypopt_wrap(nruns, its, 0.01, 1, 'component', 'normal', xT, yT, None, 0.0); t1 = time.time()
    print('hypopt_wrap nruns=%d max_its=%d: %.2f s, theta=%s nll=%.4f' % (nruns, its, t1 - t0, th, nb.mlefcn(xT, yT, None, th, 0.0)), flush=True)
