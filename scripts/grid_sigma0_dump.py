#!/usr/bin/env python3
"""-logL of the library on the sigmaY = 0 likelihood grid of fixture g6 (256 settings, n = 94), through the one-workgroup kernel
and (GPBO_LML_SMALL=0 in the environment) the general path; for the comparison with the extended-precision values.
    python3 scripts/grid_sigma0_dump.py out.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import GP
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g6_mle_grid.npz'))
th0 = g['th0']
thetas = np.array([[a, b] for a in th0 for b in th0])
gp = GP(0)
nll, stat = gp.lml_batch(g['xTrain'], g['yTrain'], thetas, np.zeros(len(thetas)))
np.savez(sys.argv[1], nll=nll.reshape(16, 16), stat=stat.reshape(16, 16))
print('failed settings', int((stat != 0).sum()))
