"""GPU: the notebook-signature functions (what a user of the reference's notebooks would import)
against the golden vectors produced by the reference's own functions."""
import numpy as np
import pytest

from conftest import TOL_MU, TOL_VAR, golden, golden_names, relmax
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def test_gprfcn_expected_improvement_argmax_like_the_notebook_cells():
    from bayesian_optimization_amd import notebook_api as nb
    g = golden('g1_mat_mixture')
    # BayesOpt_v4.ipynb:803-806 (2-D mixture data): gprfcn -> expected_improvement -> reshape of np.diag(Cov)
    Mu, Cov, K, Ks, Kss, Lm, alpha, v = nb.gprfcn(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], 0.0)
    ei = nb.expected_improvement(Mu, Cov, 0.01)
    assert Mu.shape == (6080, 1) and alpha.shape == (94, 1) and Lm.shape == (94, 94)
    var = np.diag(Cov)
    assert relmax(Mu[:, 0], g['Mu']) <= TOL_MU and relmax(var, g['var']) <= TOL_VAR and relmax(ei, g['ei']) <= TOL_VAR
    assert int(np.argmax(ei)) == int(g['argmax'])
    assert relmax(Lm, g['L']) <= 1e-9 and relmax(alpha[:, 0], g['alpha']) <= 1e-8
    Var_rs = np.reshape(np.diag(Cov), (80, 76))
    assert Var_rs.shape == (80, 76) and np.array_equal(Cov.diagonal(), var)
    with pytest.raises(TypeError):
        np.asarray(Cov)                                  # the dense m x m covariance is never formed
    # the returned-but-unused matrices materialise on demand and equal theta1 * kernel
    Kd = np.asarray(K)
    st = o.fit(g['xTrain'], g['yTrain'], g['theta'], 0.0)
    assert relmax(Kd, st['theta1'] * o.kernel_matrix(g['xTrain'], g['xTrain'], g['theta'])) <= 1e-12
    # expected_improvement on arbitrary arrays (dense covariance matrix input, as the reference's signature allows)
    mu_s = g['Mu'][:50]
    C = np.diag(g['var'][:50])
    ei_s = nb.expected_improvement(mu_s[:, None], C, 0.01)
    assert relmax(ei_s, o.expected_improvement(mu_s, g['var'][:50], 0.01)) <= 1e-12


def test_bo_loop_shape_of_the_1d_notebook_cell():
    """BayesOpt_v4.ipynb:642-664 with theta held fixed (the optimiser is out of scope): same sampled points as the oracle.
    The notebook's equispaced grid puts pairs of candidates symmetrically around a training point, which makes their EI
    equal up to the last bit (an argmax coin-flip between any two evaluation orders); jittered abscissae avoid that."""
    from bayesian_optimization_amd import notebook_api as nb
    P = 400
    x = np.sort(-5 + 10 * o.splitmix64_uniform(12, P))[:, np.newaxis]
    y = np.exp(-x ** 2) + np.sin(2 * x) * np.cos(10 * x)
    ii = [37, 180, 333]
    xT, yT = x[ii], y[ii]
    xo, yo = xT.copy(), yT.copy()
    theta = np.array([0.3])
    for it in range(6):
        Mu, Cov, _, _, _, _, _, _ = nb.gprfcn(x, xT, yT, theta, 1e-2)
        ei = nb.expected_improvement(Mu, Cov, 0.01)
        jj = int(np.argmax(ei))
        mu_o, var_o, _ = o.gprfcn_diag(x, xo, yo, theta, 1e-2)
        jo = int(np.argmax(o.expected_improvement(mu_o, var_o, 0.01)))
        assert jj == jo
        xT = np.vstack((xT, x[jj, :])); yT = np.vstack((yT, y[jj]))
        xo = np.vstack((xo, x[jo, :])); yo = np.vstack((yo, y[jo]))


def test_mlefcn_distmat_kernfcn():
    from bayesian_optimization_amd import notebook_api as nb
    g = golden('g6_mle_grid')
    xT, yT, th0 = g['xTrain'], g['yTrain'], g['th0']
    Daa = nb.distmat_v2(xT, xT)
    assert len(Daa) == 2 and np.allclose(Daa[1], np.abs(xT[:, 1][:, None] - xT[:, 1][None, :]))
    for (i, j) in ((0, 1), (1, 0), (2, 2), (5, 9)):
        v = nb.mlefcn(xT, yT, Daa, np.array([th0[i], th0[j]]), 0.05)          # the call at BayesOpt_v4.ipynb:731
        assert abs(v - g['nll_sigma005'][i, j]) <= 1e-10 * abs(g['nll_sigma005'][i, j])
    K = nb.kernfcn_v2(Daa, np.array([0.3, 0.2]))
    assert relmax(K, o.kernel_matrix(xT, xT, np.array([0.3, 0.2]))) <= 1e-14
    plain = [np.abs(xT[:, i][:, None] - xT[:7, i][None, :]) for i in range(2)]      # a plain list, as the reference's distmat_v2 builds
    K2 = nb.kernfcn_v2(plain, np.array([0.3, 0.2]))
    assert K2.shape == (94, 7) and relmax(K2, o.kernel_matrix(xT, xT[:7], np.array([0.3, 0.2]))) <= 1e-14


@pytest.mark.parametrize('name', golden_names('g7'))
def test_kriging_against_reference(name):
    from bayesian_optimization_amd import notebook_api as nb
    g = golden(name)
    deg = int(g['deg'])
    beta, cov, K, Ks, Kss, FT, FS = nb.gpr_poly(g['xT'], g['yT'], g['xT'], g['theta'], 'gaussian', deg)
    assert relmax(beta, np.atleast_1d(np.squeeze(g['beta']))) <= 1e-8
    assert abs(cov - float(g['cov'])) <= 1e-8 * abs(float(g['cov']))
    mu, mse = nb.y_kriging_v2(g['xS'], beta, cov, K, FT, g['xT'], g['yT'], g['theta'], deg, 'gaussian')
    assert relmax(mu, g['mu']) <= 1e-8
    assert np.max(np.abs(mse - g['mse'])) <= 1e-8 * max(np.max(np.abs(g['mse'])), cov ** 2)
    # one point at a time, as the notebook loops do (:521-525)
    u, v = nb.y_kriging_v2(g['xS'][3][None, :], beta, cov, K, FT, g['xT'], g['yT'], g['theta'], deg, 'gaussian')
    assert abs(u[0] - g['mu'][3]) <= 1e-8 * np.max(np.abs(g['mu']))
    assert FT.shape[0] == len(g['xT']) and FT.shape[1] == len(beta)
    if 'pi' in g.files:
        pi = nb.acquisition(g['mu'], g['yT'], np.sqrt(np.abs(g['mse'])) + 1e-300)
        assert np.max(np.abs(pi - g['pi'])) <= 1e-12


def test_multistart_hyperparameter_fit_follows_the_reference_update_rule():
    """hypopt_wrap (BayesOpt_v4.ipynb:167-186) with the notebook's settings (component-normalised sign steps, beta = 1):
    the batched device optimiser must walk the same path as the same rule driven by the oracle's -logL and gradient."""
    from bayesian_optimization_amd import notebook_api as nb
    g = golden('g1_mat_mixture')
    xT, yT = g['xTrain'], g['yTrain']
    nruns, max_its, alpha, beta = 6, 40, 0.01, 1
    x0 = 0.05 + 0.9 * o.splitmix64_uniform(21, nruns * 2).reshape(nruns, 2)
    w_hist, c_hist, flags = nb.hypopt(nruns, max_its, alpha, beta, 'component', 'normal', xT, yT, None, 0.05, x0=x0)
    for r in range(nruns):
        w = x0[r].copy()
        for k in range(max_its):
            w = w - alpha * beta * np.sign(o.mle_grad(xT, yT, w, 0.05)) * np.sqrt(2)
        assert np.max(np.abs(w_hist[r][-1] - w)) <= 1e-12
        assert abs(c_hist[r][-1] - o.mlefcn(xT, yT, w, 0.05)) <= 1e-9 * abs(c_hist[r][-1])
        assert c_hist[r][-1] < c_hist[r][0]
    theta = nb.hypopt_wrap(nruns, max_its, alpha, beta, 'component', 'normal', xT, yT, None, 0.05, x0=x0)
    best = int(np.argmin([c[-1] for c in c_hist]))
    assert np.array_equal(theta, w_hist[best][-1])


@pytest.mark.parametrize('alpha0', [0.05, 0.4])
def test_backtracking_step_rule_follows_the_reference_loop(alpha0):
    """steprule='backtrack' (BayesOpt_v4.ipynb:62-71), restated here start by start on the oracle's -logL and gradient,
    against the batched device optimiser."""
    from bayesian_optimization_amd import notebook_api as nb
    g = golden('g1_mat_mixture')
    xT, yT = g['xTrain'], g['yTrain']
    nruns, max_its, beta, sY = 4, 6, 1, 0.05
    shrunk = 0
    x0 = 0.08 + 0.5 * o.splitmix64_uniform(33, nruns * 2).reshape(nruns, 2)
    w_hist, c_hist, flags = nb.hypopt(nruns, max_its, alpha0, beta, 'full', 'backtrack', xT, yT, None, sY, x0=x0)
    gfun = lambda w: o.mlefcn(xT, yT, w, sY)
    grad = lambda w: o.mle_grad(xT, yT, w, sY)
    for r in range(nruns):
        w0, alpha, dtemp = x0[r].copy(), alpha0, 0.0
        for k in range(max_its):
            ge = grad(w0)
            ge = ge / np.linalg.norm(ge)
            dd = (1 - beta) * dtemp + beta * ge
            wtemp = w0
            w0 = w0 - alpha * dd
            h = gfun(wtemp) + grad(wtemp) @ (w0 - wtemp) + 1 / (2 * alpha) * np.linalg.norm(w0 - wtemp) ** 2
            while gfun(w0) > h:
                shrunk += 1
                alpha = 0.8 * alpha
                w0 = w0 - alpha * dd
                h = gfun(wtemp) + grad(wtemp) @ (w0 - wtemp) + 1 / (2 * alpha) * np.linalg.norm(w0 - wtemp) ** 2
                wtemp = w0
            w0 = w0 - alpha * dd
            dtemp = dd
        assert len(w_hist[r]) == max_its + 1
        assert np.max(np.abs(w_hist[r][-1] - w0)) <= 1e-9, (r, w_hist[r][-1], w0)
        assert abs(c_hist[r][-1] - gfun(w0)) <= 1e-8 * max(1.0, abs(c_hist[r][-1]))
    if alpha0 > 0.1:
        assert shrunk > 0          # the large first step really exercises the inner loop


def test_continuous_acquisition_maximisation_improves_on_the_grid(gp0):
    """Grid sweep -> top-k seeds -> batched compass search (the fminsearch step of GP/BayesOpt_v2b.m:41-46):
    the refined UCB / EI is at least the grid maximum and agrees with the oracle evaluated at the refined point."""
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import acqopt
    n, d = 60, 2
    X = o.splitmix64_uniform(31, n * d).reshape(n, d)
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + X[:, 0]
    theta = np.array([0.25, 0.3])
    g1 = np.linspace(0, 1, 40)
    grid = np.array([[a, b] for a in g1 for b in g1])
    gp0.fit(X, y, theta, 0.02)
    gp0.set_candidates(grid)
    mu_max = gp0.posterior()
    for acq, kappa in ((bo.ACQ_UCB, 1.0), (bo.ACQ_EI, 0.01)):
        idx, best = gp0.acquire(acq, kappa, bo.INCUMBENT_GIVEN, mu_max)
        vals = gp0.read(mu=False, var=False, acq=True)
        seeds = grid[acqopt.top_k(vals, 8)]
        assert np.array_equal(seeds[0], grid[idx])
        x, v, allx, allv = acqopt.maximize_acquisition(gp0, seeds, 0.0, 1.0, acq, kappa, incumbent=mu_max)
        assert v >= best and np.all((x >= 0) & (x <= 1))
        mu_o, var_o, _ = o.gprfcn_diag(x[None, :], X, y, theta, 0.02)
        ref = o.ucb(mu_o, var_o, kappa) if acq == bo.ACQ_UCB else o.expected_improvement(mu_o, var_o, kappa, incumbent=mu_max)
        assert abs(float(np.squeeze(ref)) - v) <= 1e-8 * max(abs(v), 1e-12)
        gp0.set_candidates(grid)          # restore the grid as the resident set for the next acquisition
        gp0.posterior()


def test_outer_loop_refits_sweeps_and_appends_like_the_1d_notebook_cell():
    """BayesOpt_v4.ipynb:618-664 end to end: per iteration the multi-start length-scale fit, the candidate sweep, EI, the
    first-occurrence argmax and the append.  Each iteration's pick is checked against the oracle evaluated at the theta the
    loop fitted on the training set it had at that point; the fitted theta itself against the same rule walked on the oracle."""
    from bayesian_optimization_amd.loop import bayes_opt
    P = 400
    x = np.sort(-5 + 10 * o.splitmix64_uniform(12, P))[:, np.newaxis]
    y = np.exp(-x ** 2) + np.sin(2 * x) * np.cos(10 * x)
    ii = [37, 180, 333]
    nruns, max_its, alpha = 4, 25, 0.01
    starts = lambda k, d: 0.2 + 0.6 * o.splitmix64_uniform(100 + k, nruns * d).reshape(nruns, d)
    seen = []

    def check(k, s):
        xT, yT, theta = s['xT'], s['yT'], s['theta']
        finals, costs = [], []
        for w in starts(k, 1):
            w = w.copy()
            for _ in range(max_its):
                w = w - alpha * np.sign(o.mle_grad(xT, yT, w, 1e-2))
            finals.append(w)
            costs.append(o.mlefcn(xT, yT, w, 1e-2))
        assert np.max(np.abs(theta - finals[int(np.argmin(costs))])) <= 1e-12
        mu_o, var_o, _ = o.gprfcn_diag(x, xT, yT, theta, 1e-2)
        ei_o = o.expected_improvement(mu_o, var_o, 0.01)
        assert s['jj'] == int(np.argmax(ei_o))
        assert abs(s['ei_max'] - ei_o.max()) <= 1e-8 * max(abs(ei_o.max()), 1e-3)
        seen.append(xT.shape[0])

    out = bayes_opt(x, y, x[ii], y[ii], sigmaY=1e-2, kappa=0.01, ei_tol=1e-9, nruns=nruns, max_its=max_its, alpha=alpha, beta=1,
                    normrule='component', steprule='normal', max_iter=5, starts=starts, callback=check, keep_history=True)
    assert seen == [3, 4, 5, 6, 7] and out['exitflag'] == 0
    assert out['xT'].shape == (8, 1) and out['yT'].shape == (8, 1)
    assert np.array_equal(out['xT'][3:, 0], x[out['jj_hist'], 0]) and np.array_equal(out['yT'][3:, 0], y[out['jj_hist'], 0])
    assert len(out['Mu_hist']) == 5 and out['EI_hist'][2].shape == (P,)
    assert int(np.argmax(out['EI_hist'][4])) == out['jj_hist'][4]
    # stops as soon as max EI drops under the tolerance (:663-664); a callable objective is evaluated at the picked row only
    calls = []
    f = lambda row: (calls.append(row.copy()), np.exp(-row[0] ** 2) + np.sin(2 * row[0]) * np.cos(10 * row[0]))[1]
    out2 = bayes_opt(x, f, x[ii], y[ii], sigmaY=1e-2, ei_tol=10.0, nruns=nruns, max_its=5, starts=starts)
    assert out2['exitflag'] == 1 and len(out2['jj_hist']) == 1 and len(calls) == 1 and out2['xT'].shape == (4, 1)
    assert out2['yT'][3, 0] == y[out2['jj_hist'][0], 0]


@pytest.mark.parametrize('name', golden_names('g8'))
def test_lattice_loop_on_the_notebooks_own_grid(name):
    """BayesOpt_v4.ipynb:642-664 on x = np.linspace(-5, 5, 400) (:248), theta fixed, ten iterations recorded from the reference.
    Following the reference's trajectory: the EI array agrees, the device argmax is one of the reference's numerically maximal
    candidates (mirror images about a training point tie to an ulp; the reference's pick among them is round-off, see the
    CPU test of the same fixture), and the tie-tolerant pick equals the reference's tie-tolerant pick at every iteration.
    Free-running with tie_rtol the device loop and the oracle loop pick the same ten points."""
    from bayesian_optimization_amd import notebook_api as nb
    g = golden(name)
    x, y, s, theta = g['x'], g['y'], float(g['sigmaY']), g['theta']
    ii = list(g['start'])
    xT, yT = x[ii], y[ii]
    gp = nb._gp('main')
    for it in range(len(g['jj'])):
        Mu, Cov, _, _, _, _, _, _ = nb.gprfcn(x, xT, yT, theta, s)
        ei = nb.expected_improvement(Mu, Cov, 0.01)
        ref = g['EI'][it]
        assert relmax(Mu[:, 0], g['Mu'][it]) <= 1e-9 and relmax(ei, ref) <= 1e-7
        jj = int(np.argmax(ei))
        assert ref[jj] >= ref.max() * (1 - 1e-9)
        jt, best = gp.acquire(0, 0.01, tie_rtol=1e-9)
        assert jt == o.argmax_tol(ref, 1e-9) and best == ei.max()
        jr = int(g['jj'][it])
        xT = np.vstack((xT, x[jr, :])); yT = np.vstack((yT, y[jr]))
    # free-running, tie-tolerant: device loop == oracle loop
    xT, yT = x[ii], y[ii]
    xo, yo = xT.copy(), yT.copy()
    for it in range(10):
        gp.fit(xT, yT, theta, s)
        gp.set_candidates(x)
        gp.posterior()
        jj, _ = gp.acquire(0, 0.01, tie_rtol=1e-9)
        mu_o, var_o, _ = o.gprfcn_diag(x, xo, yo, theta, s)
        jo = o.argmax_tol(o.expected_improvement(mu_o, var_o, 0.01), 1e-9)
        assert jj == jo, (it, jj, jo)
        xT = np.vstack((xT, x[jj, :])); yT = np.vstack((yT, y[jj]))
        xo = np.vstack((xo, x[jo, :])); yo = np.vstack((yo, y[jo]))


def test_first_at_least_and_exact_ties_across_blocks(gp0):
    import bayesian_optimization_amd as bo
    m = 700000
    mu = np.sin(np.arange(m) * 0.001)
    mu[[650123, 5, 300000]] = 2.0                     # three exact ties, far apart (different reduction workgroups)
    gp0.set_posterior(mu, np.ones(m))
    idx, best = gp0.acquire(bo.ACQ_UCB, 0.0)
    assert (idx, best) == (5, 2.0)
    assert gp0.first_at_least(2.0) == 5 and gp0.first_at_least(2.0 + 1e-12) == -1
    assert gp0.first_at_least(1.0 - 1e-7) == int(np.flatnonzero(mu >= 1.0 - 1e-7)[0])
    mu[5] = 2.0 - 1e-13                               # a near tie: exact argmax moves on, the tolerant one stays
    gp0.set_posterior(mu, np.ones(m))
    assert gp0.acquire(bo.ACQ_UCB, 0.0)[0] == 300000
    assert gp0.acquire(bo.ACQ_UCB, 0.0, tie_rtol=1e-9)[0] == 5


from test_oracle_golden import DESCENT_CASES, check_descent_case   # noqa: E402


@pytest.mark.parametrize('tag,norm,step', DESCENT_CASES)
def test_hypopt_walks_the_reference_descent_paths(tag, norm, step):
    """hypopt on the device-evaluated -logL + gradient against the paths the reference's own conj_gradient_descent walked
    (BayesOpt_v4.ipynb:31-87, fixture g9): component / full / unnormalised directions, momentum, constant, backtracking and
    diminishing steps."""
    from bayesian_optimization_amd import notebook_api as nb
    check_descent_case(nb.hypopt, golden('g9_descent_paths'), tag, norm, step, 1e-7)


def test_singular_kriging_matrix_returns_numbers_like_the_reference_does():
    """No jitter in this flavour (Gaussian_Process_Regression.ipynb:199-202).  On the fixture's K (cond ~ 1e19) the Cholesky
    factorisation meets a non-positive pivot; the reference's LU solves return numbers (a NEGATIVE process variance among
    them), and so does the library's LU fallback.  At this condition number nothing beyond "finite" can be asserted: the
    reference's own values change completely under one-ulp changes of K."""
    from bayesian_optimization_amd import notebook_api as nb
    g = golden('x7_kriging_singular')
    assert float(g['cov']) < 0 and float(g['condK']) > 1e16
    beta, cov, K, Ks, Kss, FT, FS = nb.gpr_poly(g['xT'], g['yT'], g['xT'], g['theta'], 'gaussian', int(g['deg']))
    mu, mse = nb.y_kriging_v2(g['xS'], beta, cov, K, FT, g['xT'], g['yT'], g['theta'], int(g['deg']), 'gaussian')
    assert np.all(np.isfinite(beta)) and np.isfinite(cov) and np.all(np.isfinite(mu)) and np.all(np.isfinite(mse))


def _cond_tol(g, base):
    return max(base, 30.0 * float(g['condK']) * 2.2e-16)


@pytest.mark.parametrize('name', golden_names('gprnb'))
def test_gpr_notebook_cells(name):
    """The GPR notebook's own driver-cell inputs (1-D nT = 3, :372-445; 2-D every nsamp-th grid point, :551-694; 3-D 800
    grid points, :811-874) through gpr_poly / y_kriging_v2, against what the reference's functions returned.  Tolerances
    grow with cond(K) * eps: K carries no jitter.  The 800-point matrix (cond 4e19) is not positive definite numerically:
    the LU fallback must return the interpolant the reference's LU returns, to the accuracy that survives."""
    from bayesian_optimization_amd import notebook_api as nb
    g = golden(name)
    deg = int(g['deg'])
    beta, cov, K, Ks, Kss, FT, FS = nb.gpr_poly(g['xT'], g['yT'], g['xT'], g['theta'], 'gaussian', deg)
    mu, mse = nb.y_kriging_v2(g['xS'], beta, cov, K, FT, g['xT'], g['yT'], g['theta'], deg, 'gaussian')
    ref_beta = np.squeeze(g['beta']).reshape(-1)
    if float(g['condK']) > 1e15:
        assert np.all(np.isfinite(mu)) and np.all(np.isfinite(mse)) and np.isfinite(cov)
        assert relmax(mu, g['mu']) <= 1e-3                       # the interpolant survives (the oracle's own LU: 5e-5)
        assert 0.5 * float(g['cov']) <= cov <= 2.0 * float(g['cov'])      # same sign and size (the oracle's own LU: 6e-3 off, this one 15 %)
        assert np.max(np.abs(mse)) <= 1e-6 * cov ** 2            # noise around zero on both sides
        return
    assert relmax(beta, ref_beta) <= _cond_tol(g, 1e-9)
    assert abs(cov - float(g['cov'])) <= _cond_tol(g, 1e-10) * abs(float(g['cov']))
    assert relmax(mu, g['mu']) <= _cond_tol(g, 1e-10)
    assert np.max(np.abs(mse - g['mse'])) <= _cond_tol(g, 1e-10) * float(g['cov']) ** 2
    if 'mle' in g.files:                                         # gpr_mle (:255-263) on the fixture's theta grid
        vals = np.array([nb.gpr_mle(np.array([t]), g['xT'], g['yT'], g['xT'], 'gaussian', deg, float(g['mle_eps'])) for t in g['mle_thetas']])
        assert np.max(np.abs(vals - g['mle']) / np.abs(g['mle'])) <= 1e-8
    if 'w0' in g.files:                                          # y_kriging / mse_kriging: r = exp(-d^2 / (2 w0)) against gpr_poly's K
        mu_w, mse_w = nb.y_kriging(g['xS'], beta, K, FT, g['xT'], g['yT'], g['w0'], deg, 'gaussian')
        assert relmax(mu_w, g['mu_w']) <= 1e-10 and np.max(np.abs(mse_w - g['mse_w'])) <= 1e-10 * cov ** 2
        m2 = nb.mse_kriging(g['xS'][7][None, :], cov, K, FT, g['xT'], g['yT'], g['w0'], deg, 'gaussian')
        assert abs(m2[0] - g['mse_w'][7]) <= 1e-10 * cov ** 2
        # and y_kriging_v2 afterwards is back on gpr_poly's own theta
        mu2, _ = nb.y_kriging_v2(g['xS'], beta, cov, K, FT, g['xT'], g['yT'], g['theta'], deg, 'gaussian')
        assert relmax(mu2, g['mu']) <= 1e-10


@pytest.mark.parametrize('name', ['gprnb_1d_theta1', 'gprnb_2d_nsamp20'])
def test_gpr_notebook_descent_on_gpr_mle(name):
    """The notebook's hyper-parameter fit (:473-510, :621-660): its conj_gradient_descent on gpr_mle, all starts in lockstep on
    device-evaluated values, against the path the reference's own loop walked (fixture: autograd's derivative replaced by a
    central difference of the reference's gpr_mle, as here)."""
    from bayesian_optimization_amd import notebook_api as nb
    from test_oracle_golden import descent_starts_with_a_slope
    g = golden(name)
    alpha, beta, its = g['descent_args']
    wh, ch = nb.gpr_descent(g['xT'], g['yT'], int(g['deg']), float(g['mle_eps']), float(alpha), float(beta), int(its), g['descent_w0'])
    rows = descent_starts_with_a_slope(g)
    assert rows
    for r in rows:
        assert len(wh[r]) == int(its)
        assert np.max(np.abs(np.squeeze(np.array(wh[r])) - g['descent_w'][r])) <= 1e-9
        assert np.max(np.abs(np.array(ch[r]) - g['descent_cost'][r]) / np.abs(g['descent_cost'][r])) <= 1e-7


def test_lu_fallback_equals_the_cholesky_route_where_both_work(gp0):
    """300 of the notebook's 3-D points through both routes: a short length scale (cond ~ 1e6: Cholesky) and a long one
    (cond > 1e19: Cholesky meets a non-positive pivot, the blocked LU with partial pivoting over three panels answers), each
    against the oracle's LU solves."""
    g = golden('gprnb_3d_n800')
    xT, yT, xS = g['xT'][:300], g['yT'][:300], g['xS']
    # theta = 0.25: cond ~ 1e6, Cholesky succeeds
    b0, c0 = gp0.kriging_fit(xT, yT, 0.25, 1)
    gp0.set_candidates(xS)
    gp0.posterior()
    mu0, mse0 = gp0.read()
    kst = o.kriging_fit(xT, yT, 0.25, 1)
    mu_o, mse_o = o.kriging_predict(kst, xS)
    assert relmax(b0, kst['beta']) <= 1e-8 and relmax(mu0, mu_o) <= 1e-9 and np.max(np.abs(mse0 - mse_o)) <= 1e-9 * c0 ** 2
    # theta = 2: cond beyond 1e19 at 300 points, Cholesky fails, LU answers; the interpolant agrees with the oracle's LU
    b1, c1 = gp0.kriging_fit(xT, yT, 2.0, 1)
    gp0.set_candidates(xS)
    gp0.posterior()
    mu1, mse1 = gp0.read()
    kst = o.kriging_fit(xT, yT, 2.0, 1)
    mu_o, mse_o = o.kriging_predict(kst, xS)
    assert np.all(np.isfinite(mu1)) and np.all(np.isfinite(mse1)) and np.isfinite(c1) and relmax(mu1, mu_o) <= 1e-2
    with pytest.raises(ValueError):
        gp0.factor()                                             # no Cholesky factor to hand out on this route


@pytest.mark.parametrize('norm,step', [('component', 'normal'), ('full', 'normal'), ('none', 'normal'), ('component', 'diminishing'),
                                       ('component', 'backtrack'), ('full', 'backtrack'), ('none', 'backtrack')])
def test_device_resident_descent_equals_the_host_loop(monkeypatch, norm, step):
    """n <= 128: the descent of all starts inside one kernel launch (gpbo_descent_small) against the host loop that calls
    the likelihood sweep once per iteration: the same iterates, lengths and exit flags, bit for bit (the update rule is
    evaluated without fused multiply-adds, as NumPy does), the same costs to the last bit or two."""
    from bayesian_optimization_amd import notebook_api as nb
    rng = np.random.default_rng(5)
    for n, d in ((7, 1), (36, 2), (70, 3), (128, 2)):
        X = rng.random((n, d))
        y = np.sin(4 * X.sum(1)) + 0.05 * rng.standard_normal(n)
        x0 = rng.random((6, d)) + 0.05
        x0[0] = 1e-4          # tiny length scale: K = I + jitter, gradient 0: the start stops at once ('full': the normalised
                              # direction is 0/0, the iterate NaN, and the start is frozen there with cost +inf on both paths)
        res = []
        for dev in ('1', '0'):
            monkeypatch.setenv('GPBO_DESCENT_DEVICE', dev)
            res.append(nb.hypopt(6, 40, 0.01, 0.7, norm, step, X, y.reshape(-1, 1), None, 0.02, x0=x0.copy()))
        (wa, ca, fa), (wb, cb, fb) = res
        assert list(fa) == list(fb)
        for r in range(6):
            assert len(wa[r]) == len(wb[r]) and len(ca[r]) == len(cb[r])
            assert np.array_equal(np.array(wa[r]), np.array(wb[r]), equal_nan=True)          # the iterates: bit for bit
            np.testing.assert_allclose(np.array(ca[r]), np.array(cb[r]), rtol=1e-13, atol=1e-12)      # -logL (a sum of terms of size ~n): two compilations of one routine


def test_gprfcn_loop_downloads_no_factor_unless_somebody_looks_at_it():
    """Every caller in the reference discards L and alpha (`_` at BayesOpt_v4.ipynb:645): a loop of gprfcn calls must not pay
    an n^2 device-to-host copy per iteration.  A handle that IS kept gets its numbers before the next call overwrites the
    model on the device."""
    from bayesian_optimization_amd import notebook_api as nb
    n, d, m = 1500, 3, 2000
    X = o.splitmix64_uniform(61, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1))[:, None]
    Xs = o.splitmix64_uniform(62, m * d).reshape(m, d)
    theta = np.array([0.4, 0.5, 0.6])
    gp = nb._gp('main')
    before = getattr(gp, 'factor_downloads', 0)
    for it in range(10):
        Mu, Cov, _, _, _, _, _, _ = nb.gprfcn(Xs, X[:n - it], y[:n - it], theta, 0.05)
        ei = nb.expected_improvement(Mu, Cov, 0.01)
    assert getattr(gp, 'factor_downloads', 0) == before
    # kept handles: filled before the model they belong to is replaced, and equal to what an eager download gives
    Mu, Cov, K, Ks, Kss, L1, a1, v = nb.gprfcn(Xs, X[:200], y[:200], theta, 0.05)
    Lref, aref = gp.factor()
    Mu2, Cov2, _, _, _, L2, a2, _ = nb.gprfcn(Xs, X[:300], y[:300], theta, 0.05)
    assert L1.shape == (200, 200) and np.array_equal(np.asarray(L1), Lref) and np.array_equal(a1[:, 0], aref)
    assert L2.shape == (300, 300) and np.asarray(L2).shape == (300, 300) and (L2 @ np.ones(300)).shape == (300,)
    st = o.fit(X[:300], y[:300], theta, 0.05)
    assert relmax(np.asarray(L2), st['L2']) <= 1e-10 and relmax(np.asarray(a2)[:, 0], st['alpha']) <= 1e-9


@pytest.mark.parametrize('normalize,alpha', [(2, 0.005), (1, 0.01), (0, 1e-4)])
def test_matlab_cg_optim_on_the_device_likelihood(normalize, alpha):
    """cg_optim / cg_optim_wrap of the MATLAB scripts (GP/BayesOpt_v3.m:457-514) on the device-evaluated -logL and analytic
    gradient against the sequential CPU restatement: under the sign rule (normalize == 2) the iterates are identical as
    long as no gradient component is within round-off of zero; under the other rules they drift by the 1e-8 of the
    gradient times the step.  MATLAB cannot run here: parity with the reference is unpinned, the restatement follows the
    script's text.  Two training sizes: the one-workgroup likelihood kernel (94 points) and the tile-engine path (300)."""
    from bayesian_optimization_amd import hyperopt
    g = golden('g9_descent_paths')
    cases = [(g['xTrain'], g['yTrain'], float(g['sigmaY']), np.asarray(g['x0'], dtype=np.float64)[:3])]
    rng = np.random.default_rng(11)
    Xb = rng.random((300, 3))
    cases.append((Xb, np.sin(3 * Xb.sum(1)) + 0.1 * rng.standard_normal(300), 0.1, rng.uniform(0.3, 0.9, (3, 3))))
    niter = 56
    for X, y, sY, x0 in cases:
        xh, fh, jh, ex = hyperopt.cg_optim(X, y, sY, x0, alpha, 0.7, niter, normalize)
        for r in range(len(x0)):
            xo, fo, jo, eo = o.cg_optim(X, y, sY, x0[r], alpha, 0.7, niter, normalize)
            assert ex[r] == eo
            assert np.max(np.abs(xh[r] - xo)) <= (1e-12 if normalize == 2 else 1e-7)
            assert np.max(np.abs(fh[r] - fo) / np.maximum(1.0, np.abs(fo))) <= 1e-8
            assert np.max(np.abs(jh[r] - jo)) <= 1e-6 * max(1.0, np.max(np.abs(jo)))
        got = hyperopt.cg_optim_wrap(X, y, sY, x0[0], None, None, alpha, 0.7, len(x0), niter, normalize, starts=x0[1:])
        want = o.cg_optim_wrap(X, y, sY, x0[0], x0[1:], alpha, 0.7, niter, normalize)
        assert np.max(np.abs(got[0] - want[0])) <= 1e-7 and abs(got[1] - want[1]) <= 1e-8 * max(1.0, abs(want[1]))
    with pytest.raises(np.linalg.LinAlgError):
        hyperopt.cg_optim(cases[0][0], cases[0][1], 0.0, np.array([[0.2, np.nan]]), 0.01, 1.0, 5, 2)
