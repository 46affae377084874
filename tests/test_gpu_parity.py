"""GPU: the HIP path, called through the C ABI (ctypes), against the oracle and the golden vectors.

Tolerances (max-norm relative, as the parity target is stated): mean 1e-10, variance 1e-8,
EI 1e-8, -logL 1e-10; argmax index identical.  Cases listed in conftest.ILL_CONDITIONED get a
loose bound because the reference algorithm itself is not reproducible there."""
import numpy as np
import pytest

import bayesian_optimization_amd as bo
from conftest import ILL_CONDITIONED, ILL_TOL_MU, ILL_TOL_NLL, ILL_TOL_VAR, TOL_MU, TOL_VAR, golden, golden_names, relmax
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu

GPR_CASES = golden_names('g1') + golden_names('g2') + golden_names('g3') + golden_names('g4') + golden_names('g5')


def synth(n, d, m, seed):
    X = o.splitmix64_uniform(seed, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1)) + 0.1 * (o.splitmix64_uniform(seed + 1000, n) - 0.5)
    Xs = o.splitmix64_uniform(seed + 2000, m * d).reshape(m, d)
    return X, y, Xs


def test_tile_engine_is_exact_on_integer_data(gp0):
    # asymmetric small-integer operands, all four operand layouts: any fragment-layout mistake shows as an O(1) error
    assert gp0.selftest_gemm(256) == 0.0
    assert gp0.selftest_gemm(384) == 0.0


@pytest.mark.parametrize('name', GPR_CASES)
def test_golden_vectors(gp0, name):
    g = golden(name)
    gp0.fit(g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    mu, var, ei, idx, ei_max, mu_max = gp0.predict_ei(g['xTest'], float(g['kappa']))
    loose = name in ILL_CONDITIONED
    assert relmax(mu, g['Mu']) <= (ILL_TOL_MU if loose else TOL_MU)
    assert relmax(var, g['var']) <= (ILL_TOL_VAR if loose else TOL_VAR)
    assert relmax(ei, g['ei']) <= (1e-4 if loose else TOL_VAR)
    # duplicated training rows with sigmaY = 0: the pivots of the duplicates are pure round-off (~1e-14), so
    # sum(log(diag L)) is not reproducible between two evaluation orders even though mean/variance are
    nll_tol = 1e-4 if name == 'g5_dup_train_s0' else (ILL_TOL_NLL if loose else 1e-10)
    assert abs(gp0.nll - float(g['nll'])) <= nll_tol * abs(float(g['nll']))
    assert idx == int(g['argmax'])
    assert idx == int(np.argmax(ei)) and ei_max == ei[idx] and mu_max == np.max(mu)
    if 'L' in g.files and not loose:
        Lm, al = gp0.factor()
        assert relmax(Lm, g['L']) <= 1e-9 and np.all(np.triu(Lm, 1) == 0)
        if name != 'g5_dup_train_s0':
            assert relmax(al, g['alpha']) <= 1e-8


@pytest.mark.parametrize('n,d,m', [(1, 1, 1), (2, 1, 5), (127, 2, 255), (128, 3, 256), (129, 5, 257), (300, 8, 1000),
                                   (257, 9, 513), (64, 16, 100)])
def test_ragged_shapes_against_oracle(gp0, n, d, m):
    X, y, Xs = synth(n, d, m, 7 + n)
    theta = 0.3 + 0.5 * o.splitmix64_uniform(n + d, d)
    if d > 4:
        theta = theta * 2
    gp0.fit(X, y, theta, 0.05)
    mu, var, ei, idx, _, _ = gp0.predict_ei(Xs, 0.01)
    mu_o, var_o, st = o.gprfcn_diag(Xs, X, y, theta, 0.05)
    ei_o = o.expected_improvement(mu_o, var_o, 0.01)
    assert relmax(mu, mu_o) <= TOL_MU and relmax(var, var_o) <= TOL_VAR
    assert np.max(np.abs(ei - ei_o)) <= TOL_VAR * max(np.max(np.abs(ei_o)), 1e-300)
    assert abs(gp0.theta1 - st['theta1']) <= 1e-10 * st['theta1']
    assert idx == int(np.argmax(ei_o))


@pytest.mark.parametrize('kernel,variant', [(bo.KERNEL_MATERN52, bo.VARIANT_V4), (bo.KERNEL_RBF, bo.VARIANT_SINGLE),
                                            (bo.KERNEL_MATERN52, bo.VARIANT_SINGLE), (bo.KERNEL_RBF, bo.VARIANT_M3),
                                            (bo.KERNEL_MATERN52, bo.VARIANT_M3)])
def test_matern_and_single_factor_variants(gp0, kernel, variant):
    # no reference implementation for Matern-5/2: pinned against the oracle, which shares all code but the epilogue
    X, y, Xs = synth(500, 4, 3000, 21)
    theta = np.array([0.5, 0.6, 0.7, 0.8])
    gp0.fit(X, y, theta, 0.1, kernel=kernel, variant=variant)
    mu, var, ei, idx, _, _ = gp0.predict_ei(Xs, 0.01)
    mu_o, var_o, st = o.gprfcn_diag(Xs, X, y, theta, 0.1, kernel=kernel, variant=variant)
    assert relmax(mu, mu_o) <= TOL_MU and relmax(var, var_o) <= TOL_VAR
    assert abs(gp0.nll - st['nll']) <= 1e-10 * abs(st['nll'])
    assert idx == int(np.argmax(o.expected_improvement(mu_o, var_o, 0.01)))


def test_acquisition_kinds_and_incumbent_modes(gp0):
    X, y, Xs = synth(200, 2, 4000, 33)
    theta = np.array([0.3, 0.4])
    gp0.fit(X, y, theta, 0.05)
    gp0.set_candidates(Xs)
    mu_max = gp0.posterior()
    mu, var = gp0.read()
    assert mu_max == np.max(mu)
    # EI with max(yTrain) as incumbent (GP/GPFR.m:48)
    idx, best = gp0.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_Y)
    ei = gp0.read(mu=False, var=False, acq=True)
    ei_o = o.expected_improvement(mu, var, 0.01, incumbent=np.max(y))
    assert relmax(ei, ei_o) <= 1e-12 and idx == int(np.argmax(ei_o))
    # given incumbent = what a multi-rank run passes after the all-reduce
    idx2, _ = gp0.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
    idx3, _ = gp0.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
    assert idx2 == idx3
    idx, best = gp0.acquire(bo.ACQ_UCB, 2.0)
    u = gp0.read(mu=False, var=False, acq=True)
    assert relmax(u, o.ucb(mu, var, 2.0)) <= 1e-14 and idx == int(np.argmax(u))
    idx, best = gp0.acquire(bo.ACQ_PI, 0.0, bo.INCUMBENT_MAX_Y)
    pi = gp0.read(mu=False, var=False, acq=True)
    assert np.max(np.abs(pi - o.probability_of_improvement(mu, np.max(y), np.sqrt(var)))) <= 1e-12


def test_ties_resolve_to_the_lowest_index_and_zero_sigma_gives_zero_ei(gp0):
    X = o.splitmix64_uniform(5, 40).reshape(20, 2)
    y = np.sin(3 * X.sum(1))                    # noise-free and short length scales: K + 1e-15*n*I stays well conditioned
    theta = np.array([0.15, 0.15])
    Xs = o.splitmix64_uniform(77, 600).reshape(300, 2)
    Xs = np.vstack([Xs, Xs, X[:5]])            # every candidate twice -> exact ties; 5 training points as candidates
    gp0.fit(X, y, theta, 0.0)
    mu, var, ei, idx, _, _ = gp0.predict_ei(Xs, 0.01)
    assert np.array_equal(mu[:300], mu[300:600]) and np.array_equal(ei[:300], ei[300:600])
    assert idx < 300 and idx == int(np.argmax(ei))
    mu_o, var_o, _ = o.gprfcn_diag(Xs, X, y, theta, 0.0)
    # sigmaY == 0: the posterior variance at a training point is zero up to the conditioning of K + 1e-15*n*I
    assert np.all(var[600:] <= 1e-9 * gp0.theta1) and relmax(mu, mu_o) <= TOL_MU and relmax(var, var_o) <= TOL_VAR
    assert np.all(ei[var == 0] == 0)


def test_errors_map_to_the_reference_behaviour(gp0):
    X, y, Xs = synth(40, 2, 50, 3)
    Xbad = X.copy()
    Xbad[7, 1] = np.nan
    with pytest.raises(np.linalg.LinAlgError):           # a NaN pivot is reported as not positive definite (LAPACK dpotrf rule)
        gp0.fit(Xbad, y, np.array([0.5, 0.5]), 0.1)
    # numerically singular K (sigmaY = 0, long length scale): np.linalg.cholesky raises at BayesOpt_v4.ipynb:142, so do we
    for n1, th1 in ((300, 5.0), (1000, 20.0)):
        X1 = o.splitmix64_uniform(3, n1).reshape(n1, 1)
        y1 = np.sin(X1.sum(1))
        with pytest.raises(np.linalg.LinAlgError):
            o.fit(X1, y1, np.array([th1]), 0.0)
        with pytest.raises(np.linalg.LinAlgError):
            gp0.fit(X1, y1, np.array([th1]), 0.0)
        assert gp0._lib.gpbo_last_info(gp0._h) >= 1
    with pytest.raises(ValueError):
        gp0.fit(X, y, np.array([0.5, 0.0]), 0.1)
    with pytest.raises(ValueError):
        gp0.fit(X, y[:-1], np.array([0.5, 0.5]), 0.1)
    gp0.fit(X, y, np.array([0.5, 0.5]), 0.1)
    with pytest.raises(ValueError):
        gp0.set_candidates(np.zeros((4, 3)))
    with pytest.raises(ValueError):
        gp0.acquire()                                      # no posterior yet
    gp0.set_candidates(Xs)
    gp0.posterior()
    assert gp0.acquire()[0] >= 0


def test_full_size_config2_properties(gp0):
    """n=4096, d=4, RBF, 1,048,576 candidates (the benchmark workload): oracle on a subsample,
    plus size-independent properties over the whole set."""
    n, d, m = 4096, 4, 1 << 20
    X, y, Xs = synth(n, d, m, 1)
    theta = np.full(d, 0.35)
    gp0.fit(X, y, theta, 0.1)
    gp0.set_candidates(Xs)
    mu_max = gp0.posterior()
    idx, best = gp0.acquire(bo.ACQ_EI, 0.01)
    mu, var, ei = gp0.read(acq=True)
    # (1) oracle on a strided subsample + the argmax candidate
    st = o.fit(X, y, theta, 0.1)
    sub = np.unique(np.concatenate([np.arange(0, m, 509), [idx]]))
    mu_o, var_o = o.predict(st, Xs[sub])
    assert relmax(mu[sub], mu_o) <= TOL_MU and relmax(var[sub], var_o) <= TOL_VAR
    assert abs(gp0.theta1 - st['theta1']) <= 1e-10 * st['theta1'] and abs(gp0.nll - st['nll']) <= 1e-10 * abs(st['nll'])
    ei_o = o.expected_improvement(mu_o, var_o, 0.01, incumbent=mu_max)
    assert relmax(ei[sub], ei_o) <= TOL_VAR
    # (2) reductions are exact functions of the arrays they reduce
    assert mu_max == np.max(mu) and idx == int(np.argmax(ei)) and best == ei[idx]
    # (3) variance bounds: 0 <= var <= theta1 (prior variance)
    assert np.all(var >= 0) and np.all(var <= gp0.theta1 * (1 + 1e-12))
    # (4) determinism and chunk-size independence, bit for bit
    gp0.posterior()
    mu2, var2 = gp0.read()
    assert np.array_equal(mu, mu2) and np.array_equal(var, var2)
    gp0.set_chunk(4096)
    gp0.posterior()
    mu3, var3 = gp0.read()
    gp0.set_chunk(0)
    assert np.array_equal(mu, mu3) and np.array_equal(var, var3)
    # (5) candidates are independent: a permuted candidate set gives the permuted outputs, bit for bit
    perm = np.random.default_rng(0).permutation(1 << 16)
    gp0.set_candidates(Xs[:1 << 16][perm])
    gp0.posterior()
    mu4, var4 = gp0.read()
    assert np.array_equal(mu4, mu[:1 << 16][perm]) and np.array_equal(var4, var[:1 << 16][perm])


def test_model_handoff_to_a_second_context_is_bit_exact(gp0):
    """What a non-fitting rank does: allocate the state, receive the four device buffers (here: a
    device-to-device copy through the same zero-copy torch views the RCCL broadcast uses), commit."""
    import torch
    from bayesian_optimization_amd import GP
    from bayesian_optimization_amd import dist as gdist
    X, y, Xs = synth(700, 5, 3000, 55)
    theta = np.full(5, 0.7)
    gp0.fit(X, y, theta, 0.1, kernel=bo.KERNEL_MATERN52)
    src, sd = gdist.model_tensors(gp0)
    Lm, _ = gp0.factor()
    W = src[0].reshape(sd.n_pad, sd.n_pad)[:700, :700].cpu().numpy()
    assert np.max(np.abs(W @ Lm - np.eye(700))) < 1e-9 and np.all(np.triu(W, 1) == 0)   # the view really is inv(L)
    gp1 = GP(0)
    gp1.state_alloc(sd.n, sd.d, sd.kernel, sd.variant, sd.sigmaY)
    dst, _ = gdist.model_tensors(gp1)
    for a, b in zip(dst, src):
        a.copy_(b)
    torch.cuda.synchronize()
    gp1.state_commit(sd.y_max)
    r0 = gp0.predict_ei(Xs, 0.01)
    r1 = gp1.predict_ei(Xs, 0.01)
    for u, v in zip(r0, r1):
        assert np.array_equal(u, v)
    with pytest.raises(ValueError):
        gp1.factor()                      # a received model carries no Cholesky factor
    gp1.close()


def test_lml_batch_and_gradient(gp0):
    """mlefcn on a (theta1, theta2, sigma) batch against the reference's own grid values, and the
    analytic gradient (MATLAB jacmlefun text) against the oracle's restatement of it."""
    from bayesian_optimization_amd import GP
    g = golden('g6_mle_grid')
    th0 = g['th0']
    X, y = g['xTrain'], g['yTrain']
    gpl = GP(0)                                   # a context of its own: the sweep reuses the model buffers as work space
    thetas = np.array([[a, b] for a in th0[::3] for b in th0[1::4]])
    for key, s in (('nll_sigma0', 0.0), ('nll_sigma005', 0.05)):
        ref = np.array([g[key][i, j] for i in range(0, 16, 3) for j in range(1, 16, 4)])
        nll, stat = gpl.lml_batch(X, y, thetas, np.full(len(thetas), s))
        ok = stat == 0
        assert ok.sum() >= len(thetas) - 2
        if s == 0.0:
            # sigmaY = 0 with length scales >= 0.3: K + 1e-15*n*I is numerically singular and the reference's own
            # values are round-off noise at the 1e-4 level; only the well-conditioned corner pins the arithmetic
            ok &= thetas.max(axis=1) <= 0.2
            assert ok.sum() >= 1
        assert np.max(np.abs(nll[ok] - ref[ok]) / np.abs(ref[ok])) <= (1e-8 if s == 0.0 else 1e-10)
    # the whole sigmaY = 0 grid (256 settings, cond up to 1e13 and beyond) against 50-digit values of the same formula: the
    # reference's own numbers are round-off there (median 1.3e-4, worst 1.2e-3 from the exact value), so what can be asked
    # is that the library is no worse a sample of that noise (this call: the one-workgroup kernel, n = 94; the general path:
    # test_sigma0_grid_general_path_against_the_exact_values) and exact where the matrix is well conditioned
    T = golden('truth_g6_sigma0')['nll']
    grid = np.array([[a, b] for a in th0 for b in th0])
    nll, stat = gpl.lml_batch(X, y, grid, np.zeros(len(grid)))
    assert np.all(stat == 0)
    e_gpu = np.abs(nll.reshape(16, 16) - T) / np.abs(T)
    e_ref = np.abs(g['nll_sigma0'] - T) / np.abs(T)
    assert np.median(e_gpu) <= 2 * np.median(e_ref) and e_gpu.max() <= 2 * e_ref.max(), (np.median(e_gpu), e_gpu.max(), np.median(e_ref), e_ref.max())
    corner = (np.maximum.outer(th0, th0) <= 0.2)
    assert e_gpu[corner].max() <= 1e-8
    # gradient, well-conditioned settings
    pts = np.array([[0.1036165, 0.0995048], [0.3, 0.2], [0.8, 1.1], [0.1036165, 0.0995048]])
    sig = np.array([0.05, 0.05, 0.05, 0.0])
    nll, grad, stat = gpl.lml_batch(X, y, pts, sig, grad=True)
    assert np.all(stat == 0)
    for i in range(len(pts)):
        g_o = o.mle_grad(X, y, pts[i], float(sig[i]))
        assert abs(nll[i] - o.mlefcn(X, y, pts[i], float(sig[i]))) <= 1e-10 * abs(nll[i])
        assert np.max(np.abs(grad[i] - g_o)) <= 1e-8 * np.max(np.abs(g_o))
    # larger, d = 8 (two tiles per side, off-diagonal tile weights) + one failing setting in the batch
    Xb, yb, _ = synth(300, 8, 4, 91)
    thb = np.vstack([np.full(8, 0.9), np.full(8, 1.3), np.full(8, 400.0)])
    sgb = np.array([0.1, 0.02, 0.0])
    nll, grad, stat = gpl.lml_batch(Xb, yb, thb, sgb, grad=True)
    # the third setting is numerically singular: whether a pivot goes non-positive depends on round-off, but the
    # status and the NaN must agree, and a failure must not disturb the other settings of the batch
    assert list(stat[:2]) == [0, 0] and (stat[2] == 1) == bool(np.isnan(nll[2]))
    for i in range(2):
        g_o = o.mle_grad(Xb, yb, thb[i], float(sgb[i]))
        assert abs(nll[i] - o.mlefcn(Xb, yb, thb[i], float(sgb[i]))) <= 1e-10 * abs(nll[i])
        assert np.max(np.abs(grad[i] - g_o)) <= 1e-8 * np.max(np.abs(g_o))
    gpl.close()


def test_incremental_append_equals_refit(gp0):
    """theta frozen, single-factor variant: append == refit on the enlarged set (oracle and GPU)."""
    from bayesian_optimization_amd import GP
    n0, d, m, k = 120, 3, 5000, 12            # crosses the 128-row pad boundary
    X, y, Xs = synth(n0 + k, d, m, 17)
    theta = np.array([0.35, 0.45, 0.55])
    gpi = GP(0)
    gpi.reserve_appends(k)
    gpi.fit(X[:n0], y[:n0], theta, 0.05, variant=bo.VARIANT_SINGLE)
    gpi.set_candidates(Xs)
    gpi.posterior()
    st = o.fit(X[:n0], y[:n0], theta, 0.05, variant=o.VARIANT_SINGLE)
    for i in range(n0, n0 + k):
        mu_max = gpi.append(X[i], y[i])
        st = o.append_point(st, X[i], y[i])
        mu, var = gpi.read()
        assert mu_max == np.max(mu)
    ref = o.fit(X, y, theta, 0.05, variant=o.VARIANT_SINGLE)
    mu_r, var_r = o.predict(ref, Xs)
    mu_a, var_a = o.predict(st, Xs)
    assert relmax(mu, mu_r) <= TOL_MU and relmax(var, var_r) <= TOL_VAR
    assert relmax(mu, mu_a) <= TOL_MU and relmax(var, var_a) <= TOL_VAR
    assert gpi.n == n0 + k and abs(gpi.theta1 - ref['theta1']) <= 1e-10 * ref['theta1']
    assert abs(gpi.nll - ref['nll']) <= 1e-10 * abs(ref['nll'])
    idx, best = gpi.acquire(bo.ACQ_EI, 0.01)
    assert idx == int(np.argmax(o.expected_improvement(mu_r, var_r, 0.01)))
    Lm, al = gpi.factor()
    assert relmax(Lm, ref['L1']) <= 1e-10 and relmax(al, ref['alpha1']) <= 1e-8
    # a full recompute from the updated factor agrees with the incrementally updated posterior
    gpi.posterior()
    mu2, var2 = gpi.read()
    assert relmax(mu2, mu) <= 1e-12 and relmax(var2, var) <= 1e-11
    with pytest.raises(ValueError):
        gpi.append(X[0], y[0])                 # reserved rows exhausted
    gpi.close()


def test_config3_shape_identities(gp0):
    """n = 32768, d = 8, Matern-5/2 (the 64M-candidate configuration's model).  A CPU oracle at this size takes
    minutes, so parity rests on identities that hold for any n: at a training input x_i
        mu_i  = y_i - sigma^2 alpha1_i                        (K (K + s2 I)^-1 y = y - s2 alpha1)
        var_i = sigma^2 - sigma^4 [(theta1 K + s2 I)^-1]_ii    (v4 two-factor form), [M^-1]_ii = sum_r W[r][i]^2
    plus bounds, determinism, and a small-n oracle check of the same code path (test_matern_and_single_factor_variants)."""
    import torch
    from bayesian_optimization_amd import dist as gdist, synth as sy
    cfg = sy.CONFIGS['cfg3']
    X, y, theta = sy.training_set(cfg)
    s2 = cfg['sigmaY'] ** 2
    gp0.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    _, alpha = gp0.factor(want_L=False)
    a1 = alpha * gp0.theta1
    (W, _, _, _), sd = gdist.model_tensors(gp0)
    minv_diag = (W.reshape(sd.n_pad, sd.n_pad)[:, :sd.n] ** 2).sum(0).cpu().numpy()
    sel = np.arange(0, cfg['n'], 7)[:4096]
    gp0.set_candidates(X[sel])
    gp0.posterior()
    mu, var = gp0.read()
    assert np.max(np.abs(mu - (y[sel] - s2 * a1[sel]))) <= 1e-10 * np.max(np.abs(y))
    assert np.max(np.abs(var - (s2 - s2 * s2 * minv_diag[sel]))) <= 1e-8 * gp0.theta1
    Xs = sy.candidates(cfg, 0, 8192)
    gp0.set_candidates(Xs)
    m1 = gp0.posterior()
    mu1, var1 = gp0.read()
    idx, best = gp0.acquire(bo.ACQ_EI, cfg['kappa'])
    ei = gp0.read(mu=False, var=False, acq=True)
    assert np.all(var1 >= 0) and np.all(var1 <= gp0.theta1 * (1 + 1e-12)) and m1 == np.max(mu1)
    assert idx == int(np.argmax(ei)) and np.all(np.isfinite(ei)) and np.all(ei >= 0)
    gp0.posterior()
    mu2, var2 = gp0.read()
    assert np.array_equal(mu1, mu2) and np.array_equal(var1, var2)


def test_cond_estimate_matches_the_factor_diagonal(gp0):
    g = golden('g3_d4_n64_s1e-2') if 'g3_d4_n64_s1e-2' in golden_names('g3') else golden(golden_names('g3')[0])
    gp0.fit(g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    Lm, _ = gp0.factor()
    dg = np.diag(Lm)
    est = gp0.cond_estimate()
    assert abs(est - (dg.max() / dg.min()) ** 2) <= 1e-12 * est
    K = Lm @ Lm.T
    assert est <= np.linalg.cond(K) * (1 + 1e-8)          # a lower bound of the 2-norm condition number


def test_small_n_fused_likelihood_equals_the_general_path(gp0):
    """n <= 128 runs one workgroup per setting; the same settings through the general (multi-launch) path and
    through the oracle must agree, including a setting whose factorisation fails."""
    import subprocess, sys, json, os
    from conftest import ROOT
    n, d = 94, 3
    X = o.splitmix64_uniform(91, n * d).reshape(n, d)
    y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, 2]
    rng = np.random.default_rng(5)
    B = 40
    thetas = rng.uniform(0.08, 0.9, size=(B, d))
    sigmas = np.where(np.arange(B) % 3 == 0, 0.0, rng.uniform(1e-3, 0.3, size=B))
    nll, grad, stat = gp0.lml_batch(X, y, thetas, sigmas, grad=True)
    ok = 0
    for b in range(B):
        ref = o.mlefcn(X, y, thetas[b], sigmas[b])
        gref = o.mle_grad(X, y, thetas[b], sigmas[b])
        cond_ok = sigmas[b] > 0 or thetas[b].max() <= 0.2          # sigma = 0 with long length scales: numerically singular
        if stat[b] == 0 and cond_ok:
            assert abs(nll[b] - ref) <= 1e-8 * max(1.0, abs(ref)), (b, nll[b], ref)
            assert np.max(np.abs(grad[b] - gref)) <= 1e-6 * max(1.0, np.max(np.abs(gref))), (b, grad[b], gref)
            ok += 1
        assert (stat[b] == 1) == bool(np.isnan(nll[b]))
    assert ok >= B // 2
    # the general paths, in child processes (the switches are read from the environment)
    code = ('import json,sys,numpy as np; sys.path.insert(0, %r); from bayesian_optimization_amd import GP; '
            'z=np.load(sys.argv[1]); g=GP(0); r=g.lml_batch(z["X"],z["y"],z["th"],z["sg"],grad=True); '
            'np.savez(sys.argv[2], nll=r[0], grad=r[1], stat=r[2])' % ROOT)
    tmp = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(tmp, exist_ok=True)
    fin, fout = os.path.join(tmp, 'small_in.npz'), os.path.join(tmp, 'small_out.npz')
    np.savez(fin, X=X, y=y, th=thetas, sg=sigmas)
    # lockstep batched context, stream lanes with graph replay, stream lanes with plain launches
    for extra in ({}, {'GPBO_LML_LOCKSTEP': '0'}, {'GPBO_LML_LOCKSTEP': '0', 'GPBO_LML_GRAPH': '0'}):
        subprocess.run([sys.executable, '-c', code, fin, fout], env=dict(os.environ, GPBO_LML_SMALL='0', **extra), check=True, timeout=120)
        z = np.load(fout)
        good = (stat == 0) & (z['stat'] == 0) & ((sigmas > 0) | (thetas.max(1) <= 0.2))
        assert good.sum() >= B // 2
        assert np.max(np.abs(nll[good] - z['nll'][good]) / np.maximum(1.0, np.abs(nll[good]))) <= 1e-9
        assert np.max(np.abs(grad[good] - z['grad'][good])) <= 1e-7 * max(1.0, np.max(np.abs(grad[good])))


def test_sigma0_grid_general_path_against_the_exact_values():
    """The sigmaY = 0 likelihood grid of fixture g6 through the tile-engine path (GPBO_LML_SMALL=0, read from the environment:
    child process): the same truth-anchored bound as for the one-workgroup kernel."""
    import subprocess, sys, os
    from conftest import ROOT
    g, T = golden('g6_mle_grid'), golden('truth_g6_sigma0')['nll']
    out = os.path.join(ROOT, 'gpurun_out', 'grid_s0_general_test.npz')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'grid_sigma0_dump.py'), out], env=dict(os.environ, GPBO_LML_SMALL='0'),
                   check=True, timeout=120)
    z = np.load(out)
    assert np.all(z['stat'] == 0)
    e_gpu = np.abs(z['nll'] - T) / np.abs(T)
    e_ref = np.abs(g['nll_sigma0'] - T) / np.abs(T)
    assert np.median(e_gpu) <= 2 * np.median(e_ref) and e_gpu.max() <= 2 * e_ref.max(), (np.median(e_gpu), e_gpu.max())
    assert e_gpu[np.maximum.outer(g['th0'], g['th0']) <= 0.2].max() <= 1e-8


@pytest.mark.parametrize('n', [60, 300])
def test_matern_gradient_small_and_general_path(gp0, n):
    """Matern-5/2 -logL gradient (no reference counterpart; the oracle's is pinned by finite differences):
    n = 60 runs the one-workgroup kernel, n = 300 the tile-engine path."""
    d = 3
    X = o.splitmix64_uniform(23, n * d).reshape(n, d)
    y = np.cos(3 * X[:, 0]) + X[:, 1] - X[:, 2] ** 2
    thetas = np.array([[0.4, 0.6, 0.5], [0.25, 0.3, 0.9], [0.8, 0.7, 0.6]])
    sigmas = np.array([0.1, 0.05, 0.2])
    nll, grad, stat = gp0.lml_batch(X, y, thetas, sigmas, kernel=bo.KERNEL_MATERN52, grad=True)
    assert np.all(stat == 0)
    for b in range(3):
        ref = o.mlefcn(X, y, thetas[b], sigmas[b], kernel=o.MATERN52)
        gref = o.mle_grad(X, y, thetas[b], sigmas[b], kernel=o.MATERN52)
        assert abs(nll[b] - ref) <= 1e-9 * max(1.0, abs(ref))
        assert np.max(np.abs(grad[b] - gref)) <= 1e-7 * max(1.0, np.max(np.abs(gref)))


@pytest.mark.parametrize('name', ['g3_d4_n512_s0', 'g5_dup_train_s0'])
def test_badly_conditioned_models_are_no_further_from_the_truth_than_the_reference(gp0, name):
    """sigmaY = 0 (the notebooks' default) leaves K + 1e-15 n I at cond up to 4e11, where the reference's double-precision
    numbers are themselves 1e-6 away from the exact posterior for the same inputs.  Fixture truth_<name> holds that exact
    posterior (50-digit arithmetic, oracle/make_truth_fixtures.py).  Requirement: every quantity the GPU returns is no further
    from the truth than the reference's own value is (or within the stated target where the reference is closer than that).
    Measured on MI355X, cond 4e11 case: theta1 1.6e-8 (reference 3.0e-7), -logL 3e-9 (5.6e-8), mean 2.7e-7 (9.9e-7),
    variance 2.6e-8 (3.4e-7): the gated refinement of alpha against the covariance matrix itself puts the GPU closer."""
    g, t = golden(name), golden('truth_' + name)
    gp0.fit(g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    mu, var, ei, idx, _, _ = gp0.predict_ei(g['xTest'], float(g['kappa']))
    _, alpha = gp0.factor(want_L=False)
    st = o.fit(g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))      # the reference's theta1 is not stored: the restatement calls the same dpotrf
    sub = t['idx']
    relsc = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
    ref = dict(theta1=relsc(st['theta1'], t['theta1']), nll=relsc(g['nll'], t['nll']), alpha=relmax(g['alpha'], t['alpha']),
               mu=relmax(g['Mu'][sub], t['mu']), var=relmax(g['var'][sub], t['var']))
    got = dict(theta1=relsc(gp0.theta1, t['theta1']), nll=relsc(gp0.nll, t['nll']), alpha=relmax(alpha, t['alpha']),
               mu=relmax(mu[sub], t['mu']), var=relmax(var[sub], t['var']))
    target = dict(theta1=1e-10, nll=1e-10, alpha=1e-8, mu=TOL_MU, var=TOL_VAR)
    print('\n%s  distance from the exact posterior  reference: %s  GPU: %s' % (
        name, {k: '%.2e' % v for k, v in ref.items()}, {k: '%.2e' % v for k, v in got.items()}))
    for k in ref:
        assert got[k] <= max(ref[k], target[k]), (k, got[k], ref[k])
