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
# numbers are only defined to ~5e-7: the same LAPACK algorithm with another panel width moves theta1 (which scales every
# variance) by 2-6e-7 and alpha by 0.6-2.3e-6, one-ulp changes of K's entries by as much, and the reference's theta1 is
# 2.8e-7 from an extended-precision evaluation (scripts/illcond_noise_floor.py; tests/test_oracle_golden.py asserts it).
# The NumPy restatement matches the reference to 1e-8 there only because it calls the same dpotrf on the same bits.
ILL_CONDITIONED = {'g3_d4_n512_s0'}
# bounds for those cases: about three times the spread of the CPU variants above
ILL_TOL_MU, ILL_TOL_VAR, ILL_TOL_NLL = 5e-6, 2e-6, 2e-7

# mean 1e-10 / variance 1e-8 (max-norm relative) are the stated targets; EI and alpha inherit the variance bound.
TOL_MU, TOL_VAR = 1e-10, 1e-8


@pytest.fixture(scope='session')
def gp0():
    from bayesian_optimization_amd import GP
    g = GP(0)
    yield g
    g.close()
