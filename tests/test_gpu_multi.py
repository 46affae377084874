"""GPU: several contexts behind one handle in ONE process (gpbo_multi_* / MultiGP), the route the notebooks take to more than
one GPU.  The test box has one GPU, so the contexts sit side by side on device 0: block splitting, the device-to-device
hand-over of the model in slices, the host threads and the host-side reductions are all exercised; only the links differ.
Everything must equal, bit for bit, one context sweeping the whole candidate set."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bayesian_optimization_amd as bo
from conftest import ROOT
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def _data(n, d, seed):
    X = o.splitmix64_uniform(seed, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1)) + 0.1 * (o.splitmix64_uniform(seed + 1, n) - 0.5)
    return X, y


@pytest.mark.parametrize('ncontexts', [2, 4])
@pytest.mark.parametrize('kernel,variant', [(bo.KERNEL_RBF, bo.VARIANT_V4), (bo.KERNEL_MATERN52, bo.VARIANT_SINGLE)])
def test_sweep_over_several_contexts_equals_one_context(gp0, ncontexts, kernel, variant):
    n, d, m = 700, 3, 10007
    X, y = _data(n, d, 5)
    Xs = o.splitmix64_uniform(9, m * d).reshape(m, d)
    Xs[m - 5] = Xs[17]                                    # an exact tie across blocks: the lower index must win
    theta = np.array([0.4, 0.5, 0.6])
    gp0.fit(X, y, theta, 0.05, kernel=kernel, variant=variant)
    ref = gp0.predict_ei(Xs, 0.01)
    mg = bo.MultiGP([0] * ncontexts)
    try:
        mg.fit(X, y, theta, 0.05, kernel=kernel, variant=variant)
        assert (mg.theta1, mg.nll, mg.n, mg.d) == (gp0.theta1, gp0.nll, n, d)
        got = mg.predict_ei(Xs, 0.01)
        for a, b in zip(got[:3], ref[:3]):
            assert np.array_equal(a, b)
        assert got[3:] == ref[3:]                          # argmax, max EI, max(Mu): identical
        blocks = mg.blocks()
        assert blocks[0][0] == 0 and blocks[-1][1] == m and all(blocks[i][1] == blocks[i + 1][0] for i in range(ncontexts - 1))
        assert max(hi - lo for lo, hi in blocks) - min(hi - lo for lo, hi in blocks) <= 1
        ms, nbytes = mg.last_distribute()
        n_pad = 768
        assert nbytes == 8.0 * n_pad * n_pad * (ncontexts - 1) and ms > 0      # every receiver ends up with all of W
        # the other acquisitions and incumbent rules, a partial read across a block boundary, the tie-tolerant pick
        for acq, kappa, mode, inc in ((bo.ACQ_UCB, 2.0, bo.INCUMBENT_MAX_MU, 0.0), (bo.ACQ_PI, 0.0, bo.INCUMBENT_MAX_Y, 0.0),
                                      (bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, 0.3)):
            assert mg.acquire(acq, kappa, mode, inc) == gp0.acquire(acq, kappa, mode, inc)
            lo = blocks[0][1] - 7
            assert np.array_equal(mg.read(mu=False, var=False, acq=True, offset=lo, count=20),
                                  gp0.read(mu=False, var=False, acq=True, offset=lo, count=20))
        i1, b1 = gp0.acquire(bo.ACQ_EI, 0.01, tie_rtol=1e-3)
        assert mg.acquire(bo.ACQ_EI, 0.01, tie_rtol=1e-3) == (i1, b1)
        # refit with candidates resident: the blocks stay where they are
        gp0.fit(X, y, theta * 1.1, 0.05, kernel=kernel, variant=variant)
        mg.fit(X, y, theta * 1.1, 0.05, kernel=kernel, variant=variant)
        assert mg.posterior() == gp0.posterior() and mg.acquire() == gp0.acquire()
        # the factor lives on context 0
        L0, a0 = gp0.factor()
        L1, a1 = mg.factor()
        assert np.array_equal(L0, L1) and np.array_equal(a0, a1)
    finally:
        mg.close()


def test_fewer_candidates_than_contexts_and_small_models(gp0):
    X, y = _data(60, 2, 11)                                # one 128-block: some slices of the hand-over are empty
    theta = np.array([0.3, 0.4])
    gp0.fit(X, y, theta, 0.05)
    mg = bo.MultiGP([0, 0, 0, 0])
    try:
        mg.fit(X, y, theta, 0.05)
        for m in (1, 2, 3, 5):
            Xs = o.splitmix64_uniform(20 + m, m * 2).reshape(m, 2)
            ref, got = gp0.predict_ei(Xs, 0.01), mg.predict_ei(Xs, 0.01)
            assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3:] == ref[3:]
        with pytest.raises(ValueError):
            mg.read(offset=3, count=10)
    finally:
        mg.close()


def test_uploaded_posterior_with_nan_and_ties_over_blocks(gp0):
    m = 4001
    mu = np.sin(np.arange(m) * 0.01)
    var = np.full(m, 0.04)
    mu[3500] = mu[200] = 2.0                               # two equal maxima in different blocks
    mg = bo.MultiGP([0, 0, 0])
    try:
        for g in (gp0, mg):
            g.set_posterior(mu, var)
        assert mg.acquire(bo.ACQ_UCB, 1.0) == gp0.acquire(bo.ACQ_UCB, 1.0) == (200, 2.2)
        assert mg.acquire(bo.ACQ_EI, 0.01) == gp0.acquire(bo.ACQ_EI, 0.01)
        mu2 = mu.copy()
        mu2[[3900, 1400]] = np.nan                         # numpy: max is NaN, argmax the first NaN
        for g in (gp0, mg):
            g.set_posterior(mu2, var)
        i0, b0 = gp0.acquire(bo.ACQ_UCB, 1.0)
        i1, b1 = mg.acquire(bo.ACQ_UCB, 1.0)
        assert i0 == i1 == 1400 and np.isnan(b0) and np.isnan(b1)
        i1, b1 = mg.acquire(bo.ACQ_EI, 0.01)
        assert i1 == 0 and np.isnan(b1) and np.all(np.isnan(mg.read(mu=False, var=False, acq=True)))
    finally:
        mg.close()


@pytest.mark.parametrize('ncontexts', [2, 3])
def test_frozen_theta_loop_over_several_contexts_picks_the_same_points(ncontexts):
    """The loop of BayesOpt_v4.ipynb:642-664 with theta frozen (border-append, GP/BayesOptFcn_v1.m:73-83): every context
    holds the model and its block of the resident V; per iteration the max(Mu) and argmax reductions happen on the host."""
    d = 2
    g1 = np.linspace(0, 1, 61)
    lattice = np.array([[a, b] for a in g1 for b in g1])   # lexicographic order, as the notebooks' nested loops (:711-714)
    f = lambda x: float(np.sin(5 * x[0]) * np.cos(3 * x[1]) + x[0])
    X0 = o.splitmix64_uniform(4, 16 * d).reshape(16, d)
    y0 = np.array([f(x) for x in X0])
    theta = np.array([0.25, 0.25])
    one, many = bo.GP(0), bo.MultiGP([0] * ncontexts)
    try:
        picks = []
        for g in (one, many):
            g.reserve_appends(12)
            g.fit(X0, y0, theta, 1e-2, variant=bo.VARIANT_SINGLE)
            g.set_candidates(lattice)
            g.posterior()
            j, best = g.acquire(bo.ACQ_EI, 0.01)
            seq = [(j, best)]
            for it in range(10):
                x = lattice[j]
                if it % 2 == 0:
                    mm, j, best = g.append_acquire(x, f(x), bo.ACQ_EI, 0.01)
                else:
                    mm = g.append(x, f(x))
                    j, best = g.acquire(bo.ACQ_EI, 0.01)
                seq.append((j, best, mm))
            picks.append(seq)
        assert picks[0] == picks[1]
        a, b = one.read(), many.read()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert (one.n, one.theta1, one.nll) == (many.n, many.theta1, many.nll) == (26, one.theta1, one.nll)
        with pytest.raises(ValueError):                    # reserved rows used up after two more
            for _ in range(3):
                many.append(lattice[5], 0.1)
    finally:
        one.close()
        many.close()


def test_likelihood_settings_split_over_contexts(gp0):
    X, y = _data(300, 3, 21)
    B = 7
    thetas = 0.3 + 0.5 * o.splitmix64_uniform(8, B * 3).reshape(B, 3)
    sig = np.array([0.05, 0.1, 0.0, 0.2, 0.05, 0.3, 0.1])
    ref = gp0.lml_batch(X, y, thetas, sig, grad=True)
    mg = bo.MultiGP([0, 0, 0])
    try:
        got = mg.lml_batch(X, y, thetas, sig, grad=True)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.all(got[2] == 0) and np.all(np.isfinite(got[0])) and np.all(np.isfinite(got[1]))
        # the trend flavour through the same handle
        xT, yT = X[:40, :2], y[:40]
        b0, c0 = gp0.kriging_fit(xT, yT, 0.3, 1)
        b1, c1 = mg.kriging_fit(xT, yT, 0.3, 1)
        assert np.array_equal(b0, b1) and c0 == c1
        Xs = o.splitmix64_uniform(31, 1000).reshape(500, 2)
        for g in (gp0, mg):
            g.set_candidates(Xs)
            g.posterior()
        a, b = gp0.read(), mg.read()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # ... and a K that is not positive definite without jitter (long length scale): the LU route's factors and row
        # exchanges are handed over as well
        xT3 = o.splitmix64_uniform(41, 300 * 3).reshape(300, 3) * 2 - 1
        yT3 = np.sin(2 * xT3[:, 0]) + np.cos(xT3[:, 1]) + 6 * np.cos(xT3[:, 2])
        b0, c0 = gp0.kriging_fit(xT3, yT3, 2.0, 0)
        with pytest.raises(ValueError):
            gp0.factor()                                   # the LU route ran: no Cholesky factor
        b1, c1 = mg.kriging_fit(xT3, yT3, 2.0, 0)
        assert np.array_equal(b0, b1) and c0 == c1
        Xs3 = o.splitmix64_uniform(42, 700 * 3).reshape(700, 3) * 2 - 1
        for g in (gp0, mg):
            g.set_candidates(Xs3)
            g.posterior()
        a, b = gp0.read(), mg.read()
        assert np.all(np.isfinite(a[0])) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        mg.close()


def test_errors_surface_with_the_reference_exception_and_the_handle_stays_usable():
    mg = bo.MultiGP([0, 0])
    try:
        X, y = _data(50, 2, 5)
        Xbad = X.copy()
        Xbad[7, 1] = np.nan                                # a NaN pivot is reported as not positive definite (LAPACK dpotrf rule)
        with pytest.raises(np.linalg.LinAlgError):
            mg.fit(Xbad, y, np.array([0.5, 0.5]), 0.0)
        with pytest.raises(ValueError):
            mg.set_candidates(X)                           # no model
        with pytest.raises(ValueError):
            mg.append(X[0], 0.1)
        mg.fit(X, y, np.array([0.3, 0.4]), 0.05)
        with pytest.raises(ValueError):
            mg.acquire()                                   # no posterior yet
        mu, var, ei, idx, _, _ = mg.predict_ei(X, 0.01)
        mu_o, var_o, _ = o.gprfcn_diag(X, X, y, np.array([0.3, 0.4]), 0.05)
        assert np.max(np.abs(mu - mu_o)) <= 1e-10 * np.max(np.abs(mu_o)) and np.max(np.abs(var - var_o)) <= 1e-8 * np.max(var_o)
    finally:
        mg.close()


def test_notebook_functions_use_every_listed_device(tmp_path):
    """GPBO_DEVICES=0,0 puts two contexts behind gprfcn / expected_improvement / bayes_opt: the notebook cells
    (BayesOpt_v4.ipynb:642-647) run unchanged and return the numbers of the one-context run, bit for bit."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from bayesian_optimization_amd import notebook_api as nb, MultiGP, GP
from bayesian_optimization_amd import synth
n, d, m = 400, 2, 6001
X = synth.uniform(71, n * d).reshape(n, d)
y = (np.sin(3 * X.sum(1)) + 0.1 * synth.normal(72, n))[:, None]
Xs = synth.uniform(73, m * d).reshape(m, d)
theta = np.array([0.3, 0.4])
Mu, Cov, _, _, _, L, alpha, _ = nb.gprfcn(Xs, X, y, theta, 0.05)
ei = nb.expected_improvement(Mu, Cov, 0.01)
kind = type(nb._gp('main')).__name__
x1 = np.linspace(-5, 5, 400)[:, None]
y1 = np.exp(-x1 ** 2) + np.sin(2 * x1) * np.cos(10 * x1)
ii = [37, 180, 333]
out = nb.bayes_opt(x1, y1, x1[ii], y1[ii], sigmaY=1e-2, nruns=4, max_its=50, max_iter=4,
                   starts=lambda k, dd: 0.2 + 0.1 * np.arange(4)[:, None] * np.ones((4, dd)))
np.savez(sys.argv[1], Mu=Mu, var=np.diag(Cov), ei=ei, L=np.asarray(L), alpha=np.asarray(alpha), kind=kind, jj=np.array(out['jj_hist']),
         eimax=np.array(out['ei_max_hist']), theta=np.array(out['theta_hist']))
'''
    outs = []
    for k, devs in enumerate((None, '0,0', '0,0,0')):
        path = str(tmp_path / ('nb_%d.npz' % k))
        env = dict(os.environ)
        env.pop('GPBO_DEVICES', None)
        if devs:
            env['GPBO_DEVICES'] = devs
        proc = subprocess.run([sys.executable, '-c', code, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
        assert proc.returncode == 0, proc.stderr[-3000:]
        outs.append(np.load(path))
    assert str(outs[0]['kind']) == 'GP' and str(outs[1]['kind']) == 'MultiGP' and str(outs[2]['kind']) == 'MultiGP'
    for o2 in outs[1:]:
        for key in ('Mu', 'var', 'ei', 'L', 'alpha', 'jj', 'eimax', 'theta'):
            assert np.array_equal(o2[key], outs[0][key]), key
