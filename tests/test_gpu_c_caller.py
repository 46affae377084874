"""GPU: the C ABI from a C99 program (tests/c_caller/bo_body.c, built by __graft_entry__.build() with gcc against
include/gpbo.h and libgpbo.so; no Python, no ctypes in that process): fit -> posterior -> EI -> argmax and one likelihood
value, against the ctypes route on the same inputs (bit for bit: the same library, the same kernels) and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, TOL_MU, TOL_VAR, relmax
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu

CALLER = os.path.join(ROOT, 'tests', 'c_caller', 'bo_body')


def run_caller(tag, X, y, theta, Xs, sigmaY, kappa):
    tmp = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(tmp, exist_ok=True)
    fin, fout = os.path.join(tmp, 'c_caller_%s_in.bin' % tag), os.path.join(tmp, 'c_caller_%s_out.bin' % tag)
    n, d = X.shape
    m = Xs.shape[0]
    with open(fin, 'wb') as f:
        np.array([n, d, m], dtype=np.int64).tofile(f)
        np.array([sigmaY, kappa], dtype=np.float64).tofile(f)
        for a in (X, y, theta, Xs):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    proc = subprocess.run([CALLER, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    with open(fout, 'rb') as f:
        head = np.fromfile(f, dtype=np.int64, count=2)
        sc = np.fromfile(f, dtype=np.float64, count=5)
        rest = np.fromfile(f, dtype=np.float64)
    out = dict(status=int(head[0]), argmax=int(head[1]), theta1=sc[0], nll=sc[1], ei_max=sc[2], mu_max=sc[3], nll_sweep=sc[4])
    if out['status'] == 0:
        out['mu'], out['var'], out['ei'] = rest[:m], rest[m:2 * m], rest[2 * m:3 * m]
    return out


def test_c99_caller_equals_the_ctypes_route_and_the_oracle(gp0):
    assert os.path.exists(CALLER), 'tests/c_caller/bo_body is missing: run __graft_entry__.build() (gcc) first'
    n, d, m = 300, 3, 4000
    X = o.splitmix64_uniform(301, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1)) + 0.05 * (o.splitmix64_uniform(302, n) - 0.5)
    Xs = o.splitmix64_uniform(303, m * d).reshape(m, d)
    theta = np.array([0.4, 0.5, 0.6])
    c = run_caller('a', X, y, theta, Xs, 0.05, 0.01)
    assert c['status'] == 0
    gp0.fit(X, y, theta, 0.05)
    mu, var, ei, idx, ei_max, mu_max = gp0.predict_ei(Xs, 0.01)
    assert np.array_equal(c['mu'], mu) and np.array_equal(c['var'], var) and np.array_equal(c['ei'], ei)
    assert c['argmax'] == idx and c['ei_max'] == ei_max and c['mu_max'] == mu_max and c['theta1'] == gp0.theta1 and c['nll'] == gp0.nll
    mu_o, var_o, st = o.gprfcn_diag(Xs, X, y, theta, 0.05)
    ei_o = o.expected_improvement(mu_o, var_o, 0.01)
    assert relmax(c['mu'], mu_o) <= TOL_MU and relmax(c['var'], var_o) <= TOL_VAR and c['argmax'] == int(np.argmax(ei_o))
    assert abs(c['nll_sweep'] - st['nll']) <= 1e-10 * abs(st['nll']) and abs(c['nll'] - st['nll']) <= 1e-10 * abs(st['nll'])


def test_c99_caller_sees_the_not_positive_definite_status():
    """NaN in the training inputs: np.linalg.cholesky raises LinAlgError in the reference (BayesOpt_v4.ipynb:142); the C
    caller gets status 1 from gpbo_fit and writes no arrays."""
    X = o.splitmix64_uniform(311, 40 * 2).reshape(40, 2)
    X[7, 1] = np.nan
    c = run_caller('b', X, np.ones(40), np.array([0.5, 0.5]), X[:5].copy() * 0 + 0.5, 0.1, 0.01)
    assert c['status'] == 1 and 'mu' not in c
