"""GPU: acqopt.fminsearch -- the MATLAB loops' continuous acquisition step (fminsearch on -acquisition_fcn_ucb from random
starts, GP/BayesOpt_v2b.m:41-46) with every start in lockstep on the device-evaluated acquisition -- against the sequential
restatement run on the CPU oracle's posterior.  The restatement is pinned to fminsearch by the documented Rosenbrock
example (tests/test_acqopt_cpu.py); MATLAB cannot run here."""
import numpy as np
import pytest

import bayesian_optimization_amd as bo
from bayesian_optimization_amd import acqopt
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('d,acq', [(1, bo.ACQ_UCB), (2, bo.ACQ_UCB), (3, bo.ACQ_UCB), (2, bo.ACQ_EI)])
def test_fminsearch_on_the_device_acquisition_follows_the_cpu_restatement(gp0, d, acq):
    rng = np.random.default_rng(40 + d)
    n = 30
    X = rng.random((n, d))
    y = np.sin(4 * X.sum(1)) + 0.3 * np.cos(7 * X[:, 0])
    theta = np.full(d, 0.3)
    sY, kappa = 0.05, 1.5 if acq == bo.ACQ_UCB else 0.01
    gp0.fit(X, y, theta, sY)
    incumbent = float(y.max()) if acq == bo.ACQ_EI else None
    X0 = rng.random((6, d))                                   # x0 = rand(1, d) in the reference, six of them at once
    x, val, flag, it, ne = acqopt.fminsearch(gp0, X0, acq, kappa, incumbent=incumbent, tol_x=1e-8, tol_fun=1e-8)

    def fun(p):
        mu, var, _ = o.gprfcn_diag(p[None, :], X, y, theta, sY)
        if acq == bo.ACQ_UCB:
            return -float(o.ucb(mu, var, kappa)[0])
        return -float(np.atleast_1d(o.expected_improvement(mu.copy(), var, kappa, incumbent=incumbent))[0])

    same_optimum = 0
    for r in range(len(X0)):
        xs, fs, fl, its, nes = o.fminsearch(fun, X0[r], tol_x=1e-8, tol_fun=1e-8)
        # the device value at the device's point is the oracle's value there
        assert abs(-fun(x[r]) - val[r]) <= 1e-9 * max(1.0, abs(val[r]))
        assert val[r] >= -fun(X0[r]) - 1e-12                  # never worse than the start
        assert flag[r] == 1 and fl == 1
        # near the optimum the simplex compares values that differ in the last bits, so the two runs need not take the same
        # steps to the end; they must arrive at the same local maximum
        if abs(val[r] + fs) <= 1e-8 * max(1.0, abs(fs)) and np.max(np.abs(x[r] - xs)) <= 1e-5:
            same_optimum += 1
    assert same_optimum >= len(X0) - 1


def test_matlab_loop_function_runs_on_the_device_and_finds_the_maximum():
    """bayes_opt_fcn_v1 (GP/BayesOptFcn_v1.m:1-61): cg_optim_wrap for the length scales, fminsearch on the acquisition, append;
    reduced descent settings so the test stays short.  The first iteration has ONE training point: the gradient of -logL is zero,
    every descent stops at once and the returned theta is the zero row of its preallocated history (the reference's own
    behaviour), so the first proposal is the random start itself; from the second iteration on the model steers."""
    from bayesian_optimization_amd import loop
    f = lambda x: float(np.exp(-8 * np.sum((np.asarray(x) - 0.6) ** 2)))
    rng = np.random.default_rng(4)
    xT, yT, th = loop.bayes_opt_fcn_v1(f, kappa=-2.0, x0=np.array([0.1, 0.2]), maxit=9, sigmaY=0.05,
                                       paropt=dict(nruns=6, niter=80, alpha=0.02, beta=1, normalize=2), rng=rng)
    assert xT.shape == (11, 2) and yT.shape == (11, 1) and len(th) == 10
    assert np.all(th[0] == 0.0) and np.all(np.isfinite(np.array(th[1:]))) and np.all(np.array(th[2:]) != 0.0)
    assert yT.max() > 0.8 and yT.max() > yT[0, 0] + 0.5           # kappa < 0 here: Mu + 2 sigma, an exploring rule that finds the bump
