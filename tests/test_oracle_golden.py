"""CPU: the oracle restatement against the golden vectors produced by running the reference
(oracle/make_golden.py), plus identities for the parts that have no runnable reference."""
import numpy as np
import pytest

from conftest import ILL_CONDITIONED, TOL_MU, TOL_VAR, golden, golden_names, relmax
from oracle import gp_oracle as o

GPR_CASES = golden_names('g1') + golden_names('g2') + golden_names('g3') + golden_names('g4') + golden_names('g5')


@pytest.mark.parametrize('name', GPR_CASES)
def test_gprfcn_ei_against_reference(name):
    g = golden(name)
    assert int(g['raised_linalgerror']) == 0
    mu, var, st = o.gprfcn_diag(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    ei = o.expected_improvement(mu, var, float(g['kappa']))
    loose = name in ILL_CONDITIONED
    assert relmax(mu, g['Mu']) <= (1e-6 if loose else TOL_MU)
    assert relmax(var, g['var']) <= (1e-6 if loose else TOL_VAR)
    assert relmax(ei, g['ei']) <= (1e-6 if loose else TOL_VAR)
    assert abs(st['nll'] - float(g['nll'])) <= 1e-10 * abs(float(g['nll']))
    assert o.argmax_first(ei) == int(g['argmax'])
    assert np.array_equal(var == 0, g['var'] == 0)          # the < 1e-60 clamp hits the same entries
    if 'L' in g.files:
        assert relmax(st['L2'], g['L']) <= 1e-9
    if name != 'g5_dup_train_s0':                            # alpha of a numerically singular K is not reproducible
        assert relmax(st['alpha'], g['alpha']) <= (1e-4 if loose else 1e-8)


def test_mle_grid_against_reference():
    g = golden('g6_mle_grid')
    th0 = g['th0']
    for key, s in (('nll_sigma0', 0.0), ('nll_sigma005', 0.05)):
        M = np.array([[o.mlefcn(g['xTrain'], g['yTrain'], np.array([a, b]), s) for b in th0] for a in th0])
        assert np.max(np.abs(M - g[key]) / np.abs(g[key])) <= 1e-9


def test_analytic_gradient_matches_reference_finite_differences():
    # central differences of the REFERENCE's mlefcn (h = 1e-6).  With sigmaY == 0 and long length
    # scales K + 1e-15*n*I is numerically singular and the reference's own differences are noise
    # (they change by orders of magnitude with h), so only the well-conditioned points pin the formula.
    g = golden('g6_mle_grid')
    checked = 0
    for si, s in enumerate(g['fd_sigmas']):
        for pi, pt in enumerate(g['fd_points']):
            if float(s) == 0.0 and pi > 0:
                continue
            grad = o.mle_grad(g['xTrain'], g['yTrain'], pt, float(s))
            fd = g['fd_grad'][pi, :, si]
            assert np.max(np.abs(grad - fd)) <= 1e-6 * np.max(np.abs(fd))
            checked += 1
    assert checked == 4


@pytest.mark.parametrize('name', golden_names('g7'))
def test_kriging_against_reference(name):
    g = golden(name)
    kst = o.kriging_fit(g['xT'], g['yT'], g['theta'], int(g['deg']))
    mu, mse = o.kriging_predict(kst, g['xS'])
    assert relmax(kst['beta'], np.atleast_1d(np.squeeze(g['beta']))) <= 1e-9
    assert abs(kst['cov'] - float(g['cov'])) <= 1e-9 * abs(float(g['cov']))
    assert relmax(mu, g['mu']) <= 1e-9
    assert np.max(np.abs(mse - g['mse'])) <= 1e-9 * max(np.max(np.abs(g['mse'])), kst['cov'] ** 2)
    if 'pi' in g.files:
        pi = o.probability_of_improvement(mu, np.max(g['yT']), np.sqrt(np.abs(mse)) + 1e-300)
        assert np.max(np.abs(pi - g['pi'])) <= 1e-7


def test_single_factor_variant_equals_v4_without_noise_rescaling():
    # with sigmaY == 0 the two posteriors agree to O(jitter) (they differ only in where eps enters)
    g = golden('g3_d4_n64_s0')
    mu4, var4, _ = o.gprfcn_diag(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], 0.0, variant=o.VARIANT_V4)
    mu1, var1, _ = o.gprfcn_diag(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], 0.0, variant=o.VARIANT_SINGLE)
    assert relmax(mu1, mu4) <= 1e-8 and relmax(var1, var4) <= 1e-6


def test_append_equals_refit():
    rng = np.random.default_rng(0)
    X = rng.random((30, 3)); y = np.sin(X.sum(1)); Xs = rng.random((200, 3))
    th = np.array([0.4, 0.5, 0.6])
    st = o.fit(X[:25], y[:25], th, 1e-2, variant=o.VARIANT_SINGLE)
    for i in range(25, 30):
        st = o.append_point(st, X[i], y[i])
    ref = o.fit(X, y, th, 1e-2, variant=o.VARIANT_SINGLE)
    mu_a, var_a = o.predict(st, Xs)
    mu_r, var_r = o.predict(ref, Xs)
    assert relmax(mu_a, mu_r) <= 1e-10 and relmax(var_a, var_r) <= 1e-9
    assert abs(st['nll'] - ref['nll']) <= 1e-10 * abs(ref['nll'])


def test_matern_shares_everything_but_the_kernel_epilogue():
    rng = np.random.default_rng(1)
    X = rng.random((40, 2)); y = np.cos(3 * X[:, 0]) + X[:, 1]; Xs = rng.random((100, 2))
    th = np.array([0.5, 0.7])
    D = o.scaled_sqdist(Xs, X, th)
    K = o.kernel_from_sqdist(D, o.MATERN52)
    r = np.sqrt(D)
    assert np.allclose(K, (1 + np.sqrt(5) * r + 5 * r ** 2 / 3) * np.exp(-np.sqrt(5) * r), rtol=1e-15)
    mu, var, st = o.gprfcn_diag(Xs, X, y, th, 1e-2, kernel=o.MATERN52)
    # posterior interpolates the data up to the noise level and never exceeds the prior variance
    mu_t, var_t = o.predict(st, X)
    assert np.max(np.abs(mu_t - y)) < 0.05 and np.all(var <= st['theta1'] * (1 + 1e-12)) and np.all(var >= 0)


def test_ei_rules():
    mu = np.array([0.1, 0.5, 0.5, -1.0])
    var = np.array([0.04, 0.0, 0.01, 1e-70])
    ei = o.expected_improvement(mu, var, 0.01)
    assert ei[1] == 0.0                      # sigma == 0 -> ei = 0  (BayesOpt_v4.ipynb:196)
    assert np.all(np.isfinite(ei)) and o.argmax_first(ei) == 2
    assert o.argmax_first(np.array([1.0, 3.0, 3.0])) == 1   # first occurrence


def test_splitmix_shards_are_slices():
    a = o.splitmix64_uniform(9, 1000)
    b = np.concatenate([o.splitmix64_uniform(9, 400), o.splitmix64_uniform(9, 600, offset=400)])
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1 and abs(a.mean() - 0.5) < 0.05


def test_oracle_matern_gradient_against_central_differences():
    """No reference implementation of the Matern-5/2 gradient: pinned by central differences of the oracle's own mlefcn."""
    n, d = 40, 3
    X = o.splitmix64_uniform(17, n * d).reshape(n, d)
    y = np.cos(3 * X[:, 0]) + X[:, 1] - X[:, 2] ** 2
    for theta, s in ((np.array([0.4, 0.6, 0.5]), 0.1), (np.array([0.25, 0.3, 0.9]), 0.02)):
        g = o.mle_grad(X, y, theta, s, kernel=o.MATERN52)
        h = 1e-6
        for i in range(d):
            e = np.zeros(d)
            e[i] = h
            fd = (o.mlefcn(X, y, theta + e, s, kernel=o.MATERN52) - o.mlefcn(X, y, theta - e, s, kernel=o.MATERN52)) / (2 * h)
            assert abs(g[i] - fd) <= 1e-5 * max(1.0, abs(fd)), (i, g[i], fd)


@pytest.mark.parametrize('name', golden_names('g8'))
def test_lattice_loop_against_reference(name):
    """The 1-D BO loop body on the notebook's own equispaced grid, theta fixed, following the reference's trajectory.
    Mirror-image candidates about a training point tie to within an ulp or two (top2_gap ~ 1e-17 in the fixture), and
    which of them np.argmax returns is round-off: at iteration 2 of the sigmaY = 1e-2 case the reference's own arithmetic
    (LU solves) and this restatement (triangular solves) pick different mirror images.  So: the EI array itself must agree,
    the pick must be one of the reference's numerically maximal candidates, and the tie-tolerant rule must agree exactly."""
    g = golden(name)
    x, y, s, theta = g['x'], g['y'], float(g['sigmaY']), g['theta']
    ii = list(g['start'])
    xT, yT = x[ii], y[ii]
    flips = 0
    for it in range(len(g['jj'])):
        mu, var, _ = o.gprfcn_diag(x, xT, yT, theta, s)
        ei = o.expected_improvement(mu, var, 0.01)
        ref = g['EI'][it]
        assert relmax(mu, g['Mu'][it]) <= 1e-9 and relmax(ei, ref) <= 1e-7
        j = o.argmax_first(ei)
        assert ref[j] >= ref.max() * (1 - 1e-9)
        flips += int(j != int(g['jj'][it]))
        assert o.argmax_tol(ei, 1e-9) == o.argmax_tol(ref, 1e-9)
        jr = int(g['jj'][it])                      # teacher forcing: continue on the reference's own trajectory
        xT = np.vstack((xT, x[jr, :]))
        yT = np.vstack((yT, y[jr]))
    if name.endswith('s1e-2'):
        assert flips >= 1                          # the documented coin flip really is in the fixture


class _OracleLml:
    """Stands in for the GPU likelihood sweep so that the optimiser's HOST logic can be checked without a GPU."""

    def lml_batch(self, X, y, thetas, sigmas, kernel=0, grad=False):
        nll = np.array([o.mlefcn(X, y, t, float(s), kernel) for t, s in zip(thetas, sigmas)])
        stat = np.zeros(len(nll), dtype=np.int32)
        if not grad:
            return nll, stat
        return nll, np.array([o.mle_grad(X, y, t, float(s), kernel) for t, s in zip(thetas, sigmas)]), stat


DESCENT_CASES = [('component_normal', 'component', 'normal'), ('component_momentum', 'component', 'normal'),
                 ('full_backtrack', 'full', 'backtrack'), ('full_diminishing', 'full', 'diminishing'), ('plain_normal', 'none', 'normal')]


def check_descent_case(hypopt, g, tag, norm, step, wtol):
    alpha, beta, its = g[tag + '_args']
    W, Cst = g[tag + '_w'], g[tag + '_cost']
    w_hist, c_hist, flags = hypopt(len(g['x0']), int(its), float(alpha), float(beta), norm, step, g['xTrain'], g['yTrain'], None,
                                   float(g['sigmaY']), x0=g['x0'])
    for r in range(len(g['x0'])):
        assert len(w_hist[r]) == W.shape[1]
        assert np.max(np.abs(np.array(w_hist[r]) - W[r])) <= wtol, (tag, r)
        assert np.max(np.abs(np.array(c_hist[r]) - Cst[r]) / np.abs(Cst[r])) <= 1e-7, (tag, r)
        assert flags[r] == g[tag + '_flag'][r]


@pytest.mark.parametrize('tag,norm,step', DESCENT_CASES)
def test_batched_descent_rule_walks_the_reference_paths(monkeypatch, tag, norm, step):
    """conj_gradient_descent of the reference (executed by make_golden.py with the analytic gradient in place of autograd's)
    against the batched restatement of its update rules in bayesian_optimization_amd/hyperopt.py, the likelihood sweep
    replaced by the CPU restatement: host logic only."""
    from bayesian_optimization_amd import hyperopt
    monkeypatch.setitem(hyperopt._ctx, None, _OracleLml())
    check_descent_case(hyperopt.hypopt, golden('g9_descent_paths'), tag, norm, step, 1e-9)


def test_singular_kriging_fixture_shows_what_the_reference_returns():
    """K without jitter at cond ~ 1e19: the reference's LU solves return finite numbers, a NEGATIVE process variance among
    them; the Cholesky-based paths (this restatement's check below, and the GPU library) refuse the matrix instead."""
    g = golden('x7_kriging_singular')
    assert float(g['condK']) > 1e16 and float(g['cov']) < 0 and np.all(np.isfinite(g['mu']))
    K = o.kernel_matrix(g['xT'], g['xT'], np.full(1, float(g['theta'][0])))
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.cholesky(K)


def test_reference_values_of_the_ill_conditioned_case_are_defined_to_about_5e_7():
    """Basis of the loose bounds on conftest.ILL_CONDITIONED: CPU variants of the SAME algorithm (Cholesky with other panel
    widths, K perturbed by one ulp per entry) move theta1 by more than 1e-7 and alpha by more than 3e-7 each, yet never by
    more than the GPU bounds allow."""
    import sys
    from conftest import ILL_TOL_MU, ILL_TOL_VAR, ROOT
    sys.path.insert(0, ROOT + '/scripts')
    from illcond_noise_floor import noise_floor
    r = noise_floor(extended=False)
    moves = list(r['panel'].values()) + r['ulp']
    assert r['cond'] > 1e11
    assert min(t for t, _ in moves) > 1e-7 and min(a for _, a in moves) > 3e-7
    assert max(t for t, _ in moves) < ILL_TOL_VAR and max(a for _, a in moves) < ILL_TOL_MU


def test_matlab_two_factor_variant_differs_from_v4_only_through_the_unscaled_noise():
    """GP/BayesOpt_v3.m:343-360 takes alpha from the second factor: Mu = theta1 Ks0 (theta1 K0 + eps)^-1 y.  With eps -> 0 that is
    v4's mean; with sigmaY != 0 it is not (eps is not rescaled by theta1), while the variance is v4's in both cases."""
    g = golden('g3_d4_n64_s1e-1')
    mu4, var4, _ = o.gprfcn_diag(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], 0.1, variant=o.VARIANT_V4)
    mu3, var3, st = o.gprfcn_diag(g['xTest'], g['xTrain'], g['yTrain'], g['theta'], 0.1, variant=o.VARIANT_M3)
    assert np.array_equal(var3, var4) and relmax(mu3, mu4) > 1e-4
    K0 = o.kernel_matrix(g['xTrain'], g['xTrain'], g['theta'])
    Ks0 = o.kernel_matrix(g['xTest'], g['xTrain'], g['theta'])
    th1 = st['theta1']
    direct = th1 * Ks0 @ np.linalg.solve(th1 * K0 + 0.01 * np.eye(64), g['yTrain'].reshape(-1))
    assert relmax(mu3, direct) <= 1e-11
    g0 = golden('g3_d4_n64_s0')
    a, _, _ = o.gprfcn_diag(g0['xTest'], g0['xTrain'], g0['yTrain'], g0['theta'], 0.0, variant=o.VARIANT_V4)
    b, _, _ = o.gprfcn_diag(g0['xTest'], g0['xTrain'], g0['yTrain'], g0['theta'], 0.0, variant=o.VARIANT_M3)
    assert relmax(a, b) <= 1e-6


# ---- the GPR notebook's own driver-cell inputs (fixtures gprnb_*, written by oracle/make_golden.py from the reference's functions)
def cond_tol(g, base):
    """Tolerance for a quantity that goes through K^-1 without jitter: `base` where K is well conditioned, growing with
    cond(K) * eps where it is not (the reference's own numbers move by that much under one-ulp changes of K's entries)."""
    return max(base, 30.0 * float(g['condK']) * 2.2e-16)


@pytest.mark.parametrize('name', golden_names('gprnb'))
def test_gpr_notebook_cells_against_the_restatement(name):
    g = golden(name)
    if float(g['condK']) > 1e15:
        pytest.skip('cond(K) = %.1e: the reference\'s LU numbers are not reproducible by any second evaluation' % float(g['condK']))
    kst = o.kriging_fit(g['xT'], g['yT'], g['theta'], int(g['deg']))
    mu, mse = o.kriging_predict(kst, g['xS'])
    assert relmax(kst['beta'], np.squeeze(g['beta']).reshape(-1)) <= cond_tol(g, 1e-11)
    assert abs(kst['cov'] - float(g['cov'])) <= cond_tol(g, 1e-12) * abs(float(g['cov']))
    assert relmax(mu, g['mu']) <= cond_tol(g, 1e-12)
    assert np.max(np.abs(mse - g['mse'])) <= cond_tol(g, 1e-12) * float(g['cov']) ** 2
    if 'mle' in g.files:
        m = np.array([o.gpr_mle(t, g['xT'], g['yT'], int(g['deg']), float(g['mle_eps'])) for t in g['mle_thetas']])
        assert np.max(np.abs(m - g['mle']) / np.abs(g['mle'])) <= 1e-8
    if 'w0' in g.files:      # y_kriging / mse_kriging with their own parameterisation of r
        mu_w, mse_w = o.kriging_predict_w(kst, g['xS'], float(g['w0'][0]))
        assert relmax(mu_w, g['mu_w']) <= 1e-12 and np.max(np.abs(mse_w - g['mse_w'])) <= 1e-12 * float(g['cov']) ** 2


def descent_starts_with_a_slope(g):
    """Starts of a gprnb descent fixture whose path is decided by the function, not by round-off: on the flat part of gpr_mle
    (tiny theta, K = I to the last bit) the sign of a difference quotient is noise and the reference's own path is arbitrary."""
    C = g['descent_cost']
    return [r for r in range(len(g['descent_w0'])) if abs(C[r, 0] - C[r, 5]) > 1e-6 * abs(C[r, 0])]


@pytest.mark.parametrize('name', ['gprnb_1d_theta1', 'gprnb_2d_nsamp20'])
def test_gpr_notebook_descent_path_against_the_restatement(name):
    g = golden(name)
    alpha, beta, its = g['descent_args']
    deg, eps = int(g['deg']), float(g['mle_eps'])
    fun = lambda w: o.gpr_mle(w[0], g['xT'], g['yT'], deg, eps)
    dfun = lambda w: np.array([(fun(w + 1e-6 * np.abs(w)) - fun(w - 1e-6 * np.abs(w))) / (2e-6 * abs(w[0]))])
    rows = descent_starts_with_a_slope(g)
    assert rows
    for r in rows:
        wh, ch = o.gpr_descent(fun, dfun, float(alpha), float(beta), int(its), np.array([g['descent_w0'][r]]))
        assert np.max(np.abs(np.squeeze(np.array(wh)) - g['descent_w'][r])) <= 1e-9
        assert np.max(np.abs(np.array(ch) - g['descent_cost'][r]) / np.abs(g['descent_cost'][r])) <= 1e-7


@pytest.mark.parametrize('name', ['g3_d4_n512_s0', 'g5_dup_train_s0'])
def test_reference_distance_from_the_exact_posterior_at_bad_conditioning(name):
    """truth_<name>: the exact posterior for the fixture's inputs (50-digit arithmetic).  Documents how far the reference's
    own double-precision numbers are from it -- the yardstick of the GPU test for these cases -- and that the restatement
    sits at the same distance."""
    g, t = golden(name), golden('truth_' + name)
    sub = t['idx']
    mu_o, var_o, st = o.gprfcn_diag(g['xTest'][sub], g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    ref_mu, ref_var = relmax(g['Mu'][sub], t['mu']), relmax(g['var'][sub], t['var'])
    if name == 'g3_d4_n512_s0':
        assert 1e-7 < ref_mu < 5e-6 and 5e-8 < ref_var < 2e-6            # cond 4e11: the reference is 1e-6 from the truth
        assert abs(st['theta1'] - float(t['theta1'])) / float(t['theta1']) > 5e-8
    else:
        assert ref_mu < 1e-11 and ref_var < 1e-10                         # duplicated rows: mean and variance are fine ...
        assert abs(float(g['nll']) - float(t['nll'])) / abs(float(t['nll'])) > 1e-6      # ... -logL is not (pivots of pure round-off)
    assert relmax(mu_o, t['mu']) <= 2 * ref_mu + 1e-12 and relmax(var_o, t['var']) <= 2 * ref_var + 1e-12


def test_sigma0_likelihood_grid_reference_and_restatement_against_the_exact_values():
    """truth_g6_sigma0: mlefcn on the 16 x 16 length-scale grid at sigmaY = 0 in 50-digit arithmetic.  Away from the short
    length scales the reference's own values are round-off (median 1.3e-4, worst 1.2e-3 from the exact value); the
    restatement, which calls the same LAPACK routines, is the same kind of sample."""
    g, T = golden('g6_mle_grid'), golden('truth_g6_sigma0')['nll']
    th0 = g['th0']
    e_ref = np.abs(g['nll_sigma0'] - T) / np.abs(T)
    assert 5e-5 < np.median(e_ref) < 5e-4 and 3e-4 < e_ref.max() < 5e-3
    corner = np.maximum.outer(th0, th0) <= 0.2
    assert e_ref[corner].max() <= 1e-9
    mine = np.array([[o.mlefcn(g['xTrain'], g['yTrain'], np.array([a, b]), 0.0) for b in th0] for a in th0])
    e_o = np.abs(mine - T) / np.abs(T)
    assert np.median(e_o) <= 2 * np.median(e_ref) and e_o.max() <= 2 * e_ref.max()


def _cg_problem():
    g = golden('g9_descent_paths')
    return g['xTrain'], g['yTrain'], float(g['sigmaY']), np.asarray(g['x0'], dtype=np.float64)


@pytest.mark.parametrize('normalize', [0, 1, 2])
def test_matlab_cg_optim_lockstep_equals_the_sequential_restatement(monkeypatch, normalize):
    """cg_optim of the MATLAB scripts (GP/BayesOpt_v3.m:457-484): the lockstep version in hyperopt.py, run on the CPU
    restatement of the likelihood, against the line-by-line sequential restatement in the oracle -- same iterates, same
    costs, same gradients, bit for bit (host logic only).  MATLAB cannot run here: parity with the reference is unpinned,
    the restatement follows the script's text."""
    from bayesian_optimization_amd import hyperopt
    monkeypatch.setitem(hyperopt._ctx, None, _OracleLml())
    X, y, sY, x0 = _cg_problem()
    alpha = {0: 1e-4, 1: 0.01, 2: 0.005}[normalize]
    xh, fh, jh, ex = hyperopt.cg_optim(X, y, sY, x0, alpha, 0.7, 25, normalize)
    for r in range(len(x0)):
        xo, fo, jo, eo = o.cg_optim(X, y, sY, x0[r], alpha, 0.7, 25, normalize)
        assert np.array_equal(xh[r], xo) and np.array_equal(fh[r], fo) and np.array_equal(jh[r], jo) and ex[r] == eo
    assert np.all(fh[:, -1] < fh[:, 0])                       # it is a descent


def test_matlab_cg_optim_early_stop_leaves_zero_rows_and_wrap_picks_by_mean_of_the_tail(monkeypatch):
    """A start that stops early (step <= 1e-9) leaves zero rows behind its last iterate (the histories are preallocated,
    :460-462), so the wrapper's x1_hist(end,:) is zero for it and its tail mean counts the zeros (:492-493, :506-511)."""
    from bayesian_optimization_amd import hyperopt
    monkeypatch.setitem(hyperopt._ctx, None, _OracleLml())
    X, y, sY, x0 = _cg_problem()
    xh, fh, jh, ex = hyperopt.cg_optim(X, y, sY, x0[:2], 1e-13, 1.0, 6, 2)
    assert list(ex) == [1, 1] and np.all(xh[:, 2:] == 0) and np.all(fh[:, 2:] == 0) and np.all(xh[:, 1] != 0)
    niter = 60
    got = hyperopt.cg_optim_wrap(X, y, sY, x0[0], None, None, 0.004, 0.8, len(x0), niter, 2, starts=x0[1:])
    want = o.cg_optim_wrap(X, y, sY, x0[0], x0[1:], 0.004, 0.8, niter, 2)
    assert np.array_equal(got[0], want[0]) and got[1] == want[1]
    with pytest.raises(ValueError):
        hyperopt.cg_optim_wrap(X, y, sY, x0[0], None, None, 0.004, 0.8, 2, 50, 2, starts=x0[1:2])
