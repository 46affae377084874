import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def relmax(a, b):
    """max-norm relative error: max|a-b| / max|b| (the norm the parity targets are stated in)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


# Golden cases whose K + eps*I is so ill-conditioned (sigmaY = 0 -> jitter 1e-15*n, cond 4e11) that the reference's own
# numbers are 1e-6 away from the exact posterior of the same inputs (fixture truth_g3_d4_n512_s0, 50-digit arithmetic:
# reference theta1 3.0e-7, mean 9.9e-7, variance 3.4e-7 from the truth; the GPU 1.6e-8 / 2.7e-7 / 2.6e-8).  The principled
# bound is test_badly_conditioned_models_are_no_further_from_the_truth_than_the_reference (GPU no further from the truth than
# the reference is); the direct GPU-vs-reference comparison of the golden-vector test can then only be asked to hold to the
# sum of the two distances, which is what the constants below are.
ILL_CONDITIONED = {'g3_d4_n512_s0'}
ILL_TOL_MU, ILL_TOL_VAR, ILL_TOL_NLL = 3e-6, 1e-6, 2e-7

# mean 1e-10 / variance 1e-8 (max-norm relative) are the stated targets; EI and alpha inherit the variance bound.
TOL_MU, TOL_VAR = 1e-10, 1e-8


@pytest.fixture(scope='session')
def gp0():
    from bayesian_optimization_amd import GP
    g = GP(0)
    yield g
    g.close()
