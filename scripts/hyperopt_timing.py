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
