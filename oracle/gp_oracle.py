"""CPU restatement (NumPy) of the reference GP posterior / EI / likelihood path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bayesian_optimization_amd/`` imports this
module; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / timed CPU baseline, never as
the shipped compute path.

Parity status: PINNED.  The reference repository holds no tests or golden vectors
for this path, so the restatement is pinned against outputs of the reference's own
functions executed in the build container (``oracle/make_golden.py`` -> the
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks them).
The Matern-5/2 kernel, the incremental append and the analytic likelihood gradient
have no runnable reference counterpart (see the notes on each function) and are
pinned only by identities (append == refit, finite differences, formula text).

Every function cites the reference lines it restates (paths relative to the
reference tree, notebook line numbers are raw-JSON line numbers).

Differences from the reference that do not change results beyond round-off:
  * only diag(Cov) is formed (callers only consume np.diag(Cov):
    BayesOpt/BayesOpt_v4.ipynb:191,208,806), never the m x m matrices;
  * triangular systems are solved with triangular solves instead of LU
    (the reference calls np.linalg.solve on a triangular matrix, :143,:150).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import solve_triangular
from scipy.special import ndtr

RBF = 0
MATERN52 = 1

VARIANT_V4 = 0       # two-factor posterior of BayesOpt_v4.ipynb:142-151
VARIANT_SINGLE = 1   # single-factor, theta1-scaled variance of GP/BayesOptFcn_v1.m:73-83
VARIANT_M3 = 3       # MATLAB two-factor form of GP/BayesOpt_v3.m:343-360 (alpha from the second factor); text only, not runnable here


# --------------------------------------------------------------------------- kernels
def scaled_sqdist(A, B, theta):
    """D[p,q] = sum_i (A[p,i]-B[q,i])^2 / theta[i]^2.

    Restates distmat_v2 + the accumulation loop of kernfcn_v2
    (BayesOpt/BayesOpt_v4.ipynb:100-105, :119-123) without materialising the d
    per-dimension matrices.  Direct-difference form (the reference goes through
    scipy.spatial.distance_matrix, which also takes differences first).
    """
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    D = np.zeros((A.shape[0], B.shape[0]))
    for i in range(A.shape[1]):
        diff = np.abs(A[:, i][:, None] - B[:, i][None, :])
        D += diff ** 2 / theta[i] ** 2
    return D


def kernel_from_sqdist(D, kernel=RBF):
    """RBF: exp(-D/2) (BayesOpt_v4.ipynb:123).  Matern-5/2 (no reference
    implementation; closed form (1+sqrt5 r+5r^2/3) exp(-sqrt5 r), r = sqrt(D))."""
    if kernel == RBF:
        return np.exp(-1 / 2 * D)
    r = np.sqrt(D)
    s5r = math.sqrt(5.0) * r
    return (1.0 + s5r + (5.0 / 3.0) * D) * np.exp(-s5r)


def kernel_matrix(A, B, theta, kernel=RBF):
    return kernel_from_sqdist(scaled_sqdist(A, B, theta), kernel)


def jitter(n, sigmaY, variant=0):
    """eps of BayesOpt_v4.ipynb:138-141 (scalar on the diagonal).  The MATLAB two-factor form (variant 3) uses
    eps*length(K) with MATLAB's eps = 2^-52 (GP/BayesOpt_v3.m:343,352)."""
    if sigmaY == 0:
        return (2.220446049250313e-16 if variant == 3 else 1e-15) * n
    return sigmaY ** 2


# --------------------------------------------------------------------------- fit
def fit(xTrain, yTrain, theta, sigmaY, kernel=RBF, variant=VARIANT_V4):
    """Everything of gprfcn that does not depend on xTest
    (BayesOpt_v4.ipynb:134,138-148).  Raises numpy.linalg.LinAlgError exactly
    where the reference's np.linalg.cholesky does."""
    X = np.asarray(xTrain, dtype=np.float64)
    y = np.asarray(yTrain, dtype=np.float64).reshape(-1)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    n = X.shape[0]
    K0 = kernel_matrix(X, X, theta, kernel)
    eps = jitter(n, sigmaY, variant)
    Keps = K0 + eps * np.eye(n)
    L1 = np.linalg.cholesky(Keps)                                 # :142
    alpha1 = solve_triangular(L1, y, lower=True)
    alpha1 = solve_triangular(L1, alpha1, lower=True, trans='T')  # :143
    theta1 = float(y @ alpha1) / n                                # :144
    nll = n / 2 * (math.log(2 * math.pi) + 1) + n / 2 * math.log(theta1) \
        + float(np.sum(np.log(np.diag(L1))))                      # :164 (negated)
    st = dict(X=X, y=y, theta=theta, sigmaY=float(sigmaY), kernel=kernel, variant=variant,
              n=n, L1=L1, alpha1=alpha1, theta1=theta1, nll=nll, eps=eps)
    if variant == VARIANT_V4:
        L2 = np.linalg.cholesky(theta1 * K0 + eps * np.eye(n))   # :146-147
        st['L2'] = L2
        st['alpha'] = alpha1 / theta1                             # :148 (returned alpha)
    elif variant == VARIANT_M3:
        L2 = np.linalg.cholesky(theta1 * K0 + eps * np.eye(n))   # GP/BayesOpt_v3.m:350-351
        a2 = solve_triangular(L2, y, lower=True)
        st['L2'] = L2
        st['alpha'] = solve_triangular(L2, a2, lower=True, trans='T')   # GP/BayesOpt_v3.m:357: alpha = L'\(L\y) with the second L
        # :362: logL from y'alpha and diag(L) of the second factor
        st['nll'] = n / 2 * (math.log(2 * math.pi) + 1) + n / 2 * math.log(float(y @ st['alpha']) / n) \
            + float(np.sum(np.log(np.diag(L2))))
    else:
        st['L2'] = None
        st['alpha'] = alpha1
    return st


def predict(st, xTest, block=8192):
    """Posterior mean and diag(Cov) (BayesOpt_v4.ipynb:135,146,149-151).

    v4:      Mu = (theta1 Ks0)(alpha1/theta1);  var = theta1 - sum_k v_k^2,
             v = L2^{-1} (theta1 Ks0^T);  var < 1e-60 -> 0.
    single:  Mu = Ks0 alpha1;  var = theta1 (1 - sum_k v_k^2), v = L1^{-1} Ks0^T
             (GP/BayesOptFcn_v1.m:79-83), same clamp.
    """
    Xs = np.asarray(xTest, dtype=np.float64)
    m = Xs.shape[0]
    mu = np.empty(m)
    var = np.empty(m)
    th1 = st['theta1']
    for lo in range(0, m, block):
        hi = min(m, lo + block)
        Ks0 = kernel_matrix(Xs[lo:hi], st['X'], st['theta'], st['kernel'])   # (mb, n)
        if st['variant'] in (VARIANT_V4, VARIANT_M3):
            Ks = th1 * Ks0
            mu[lo:hi] = Ks @ st['alpha']
            v = solve_triangular(st['L2'], Ks.T, lower=True)
            c = th1 * 1.0 - np.sum(v * v, axis=0)
        else:
            mu[lo:hi] = Ks0 @ st['alpha1']
            v = solve_triangular(st['L1'], Ks0.T, lower=True)
            c = th1 * (1.0 - np.sum(v * v, axis=0))
        c[c < 1e-60] = 0.0                                                   # :151
        var[lo:hi] = c
    return mu, var


def gprfcn_diag(xTest, xTrain, yTrain, theta, sigmaY, kernel=RBF, variant=VARIANT_V4):
    """(Mu (m,), diag(Cov) (m,), state) for the call gprfcn(xTest,xTrain,yTrain,theta,sigmaY)
    (BayesOpt_v4.ipynb:127-152)."""
    st = fit(xTrain, yTrain, theta, sigmaY, kernel, variant)
    mu, var = predict(st, xTest)
    return mu, var, st


# --------------------------------------------------------------------------- acquisition
def expected_improvement(Mu, var, kappa, incumbent=None):
    """BayesOpt_v4.ipynb:188-197 with diag(Cov) passed directly.
    incumbent=None -> max(Mu) (the notebook's rule, :193)."""
    Mu = np.squeeze(np.asarray(Mu, dtype=np.float64))
    sigma = np.sqrt(np.asarray(var, dtype=np.float64))
    inc = np.max(Mu) if incumbent is None else incumbent
    with np.errstate(divide='ignore', invalid='ignore'):
        imp = Mu - inc - kappa
        Z = imp / sigma
        ei = imp * ndtr(Z) + sigma * (np.exp(-Z ** 2 / 2.0) / math.sqrt(2 * math.pi))
        ei[sigma == 0.0] = 0.0
    return ei


def ucb(Mu, var, kappa):
    """GP/BayesOpt_v1.m:248-251: Mu + kappa*sqrt(diag(Cov))."""
    return np.squeeze(Mu) + kappa * np.sqrt(var)


def probability_of_improvement(mu, ymax, sigma):
    """acquisition() of Gaussian_Process_Regression.ipynb:265-268."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return ndtr((np.asarray(mu) - ymax) / np.asarray(sigma))


def argmax_first(a):
    """np.argmax semantics (first occurrence), BayesOpt_v4.ipynb:647."""
    return int(np.argmax(a))


def argmax_tol(a, rtol):
    """Tie-tolerant argmax: the lowest index whose value is within rtol*|max| of the maximum.  No reference counterpart
    (np.argmax at BayesOpt_v4.ipynb:647 is rtol = 0); it makes the pick independent of the last bits of values that are
    mathematically equal, which on the notebook's equispaced grid differ between any two evaluation orders."""
    a = np.asarray(a, dtype=np.float64)
    best = np.max(a)
    return int(np.flatnonzero(a >= best - rtol * abs(best))[0])


# --------------------------------------------------------------------------- likelihood
def mlefcn(xTrain, yTrain, theta, sigmaY, kernel=RBF):
    """-logL of BayesOpt_v4.ipynb:154-165 (Daa argument folded into the kernel build)."""
    X = np.asarray(xTrain, dtype=np.float64)
    y = np.asarray(yTrain, dtype=np.float64).reshape(-1)
    n = X.shape[0]
    K = kernel_matrix(X, X, theta, kernel) + jitter(n, sigmaY) * np.eye(n)
    L = np.linalg.cholesky(K)
    a = solve_triangular(L, y, lower=True)
    a = solve_triangular(L, a, lower=True, trans='T')
    theta1 = float(y @ a) / n
    return n / 2 * (math.log(2 * math.pi) + 1) + n / 2 * math.log(theta1) + float(np.sum(np.log(np.diag(L))))


def mle_grad(xTrain, yTrain, theta, sigmaY, kernel=RBF):
    """Analytic d(-logL)/d theta_i.  Restates jacmlefun
    (GP/BayesOpt_v3.m:392-417; MATLAB text, cannot be executed here):
      dK_i = Dsq_i o (L L^T) / theta_i^3,
      J_i  = -1/2 ( n a^T dK_i a / (y^T a) - tr((L L^T)^{-1} dK_i) ).
    Matern-5/2 (no reference implementation): dK_i = Dsq_i / theta_i^3 * (5/3)(1 + s) exp(-s), s = sqrt(5 D).
    Pinned by central differences of mlefcn in the tests."""
    X = np.asarray(xTrain, dtype=np.float64)
    y = np.asarray(yTrain, dtype=np.float64).reshape(-1)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    n, d = X.shape
    K = kernel_matrix(X, X, theta, kernel) + jitter(n, sigmaY) * np.eye(n)
    L = np.linalg.cholesky(K)
    a = solve_triangular(L, y, lower=True)
    a = solve_triangular(L, a, lower=True, trans='T')
    W = solve_triangular(L, np.eye(n), lower=True)
    Kinv = W.T @ W
    g = np.empty(d)
    ya = float(y @ a)
    if kernel == RBF:
        F = K                                   # -2 dk/d(r^2) = k for the RBF kernel
    else:
        s5r = np.sqrt(5.0 * scaled_sqdist(X, X, theta))
        F = (5.0 / 3.0) * (1.0 + s5r) * np.exp(-s5r)
    for i in range(d):
        Dsq = (X[:, i][:, None] - X[:, i][None, :]) ** 2
        dK = Dsq * F / theta[i] ** 3
        g[i] = -0.5 * (n * float(a @ dK @ a) / ya - float(np.sum(Kinv * dK)))
    return g


# --------------------------------------------------------------------------- incremental append
def append_point(st, x_new, y_new):
    """Border-append one training point with theta frozen (single-factor variant
    only).  No reference implementation (the reference refits from scratch every
    iteration, BayesOpt_v4.ipynb:642-645); must equal fit() on the enlarged set
    except that the jitter keeps the value it had at the first fit when sigmaY==0.
    Returns a new state."""
    assert st['variant'] == VARIANT_SINGLE
    X, y, L = st['X'], st['y'], st['L1']
    n = st['n']
    xn = np.asarray(x_new, dtype=np.float64).reshape(1, -1)
    k = kernel_matrix(xn, X, st['theta'], st['kernel']).reshape(-1)
    l = solve_triangular(L, k, lower=True)
    dd = 1.0 + st['eps'] - float(l @ l)
    if not dd > 0:
        raise np.linalg.LinAlgError('append: matrix not positive definite')
    Ln = np.zeros((n + 1, n + 1))
    Ln[:n, :n] = L
    Ln[n, :n] = l
    Ln[n, n] = math.sqrt(dd)
    Xn = np.vstack([X, xn])
    yn = np.concatenate([y, [float(y_new)]])
    a = solve_triangular(Ln, yn, lower=True)
    a = solve_triangular(Ln, a, lower=True, trans='T')
    th1 = float(yn @ a) / (n + 1)
    out = dict(st)
    out.update(X=Xn, y=yn, n=n + 1, L1=Ln, alpha1=a, alpha=a, theta1=th1,
               nll=(n + 1) / 2 * (math.log(2 * math.pi) + 1) + (n + 1) / 2 * math.log(th1)
               + float(np.sum(np.log(np.diag(Ln)))))
    return out


# --------------------------------------------------------------------------- kriging (trend GP)
def polymat(x, deg):
    """Trend basis of Gaussian_Process_Regression.ipynb:113-140 (C = 1, 2, 3 columns)."""
    x = np.asarray(x, dtype=np.float64)
    N, C = x.shape
    one = np.ones((N, 1))
    c = [x[:, i][:, None] for i in range(C)]
    if deg == 0:
        return one
    if C == 1:
        return np.hstack([one, c[0]]) if deg == 1 else np.hstack([one, c[0], c[0] ** 2])
    if C == 2:
        if deg == 1:
            return np.hstack([one, c[0], c[1]])
        if deg == 2:
            return np.hstack([one, c[0], c[1], c[0] ** 2, c[1] ** 2])
        return np.hstack([one, c[0], c[1], c[0] ** 2, c[1] ** 2, c[0] * c[1]])
    if C == 3:
        if deg == 1:
            return np.hstack([one] + c)
        if deg == 2:
            return np.hstack([one] + c + [ci ** 2 for ci in c])
        return np.hstack([one] + c + [ci ** 2 for ci in c] + [c[0] * c[1], c[0] * c[2], c[1] * c[2]])
    raise ValueError('polymat: 1..3 input columns only (as in the reference)')


def kriging_fit(xT, yT, theta, deg):
    """gpr_poly of Gaussian_Process_Regression.ipynb:197-203: isotropic
    K = exp(-D^2/(2 theta^2)) with NO jitter, GLS trend beta, cov = r^T K^-1 r / nT."""
    xT = np.asarray(xT, dtype=np.float64)
    yT = np.asarray(yT, dtype=np.float64).reshape(-1)
    th = float(np.squeeze(theta))
    nT = xT.shape[0]
    FT = polymat(xT, deg)
    K = kernel_matrix(xT, xT, np.full(xT.shape[1], th), RBF)
    KiF = np.linalg.solve(K, FT)
    Kiy = np.linalg.solve(K, yT)
    beta = np.linalg.solve(FT.T @ KiF, FT.T @ Kiy)
    res = yT - FT @ beta
    Kir = np.linalg.solve(K, res)
    cov = float(res @ Kir) / nT
    return dict(xT=xT, yT=yT, theta=th, deg=deg, FT=FT, K=K, beta=beta, cov=cov, Kir=Kir)


def kriging_predict(kst, xS):
    """y_kriging_v2 of Gaussian_Process_Regression.ipynb:228-234 for a batch of points:
    mu = f beta + r^T K^-1 (y - F beta);  mse = cov^2 (1 - r^T K^-1 r)."""
    xS = np.asarray(xS, dtype=np.float64)
    r = kernel_matrix(kst['xT'], xS, np.full(xS.shape[1], kst['theta']), RBF)   # (nT, m)
    f = polymat(xS, kst['deg'])
    mu = f @ kst['beta'] + r.T @ kst['Kir']
    Kinv_r = np.linalg.solve(kst['K'], r)
    mse = kst['cov'] ** 2 * (1.0 - np.sum(r * Kinv_r, axis=0))
    return mu, mse


def gpr_mle(theta, xT, yT, deg, epsilon):
    """gpr_mle of Gaussian_Process_Regression.ipynb:255-263 (the quantity the notebook's descent minimises, :476, :622):
        -mle = 1/2 log det chol(K + epsilon I) + r^T r / (2 cov^2) + N/2 log(2 pi cov^2),   r = yT - FT beta,
    beta and cov from gpr_poly on the UN-jittered K (LU solves); epsilon only enters the Cholesky factor whose determinant
    is taken.  The reference takes np.linalg.det of the triangular factor (the product of its diagonal, which underflows
    for a few hundred points); here the sum of the logs of the diagonal."""
    kst = kriging_fit(xT, yT, theta, deg)
    N = kst['xT'].shape[0]
    Lc = np.linalg.cholesky(kst['K'] + epsilon * np.eye(N))
    res = kst['yT'] - kst['FT'] @ kst['beta']
    cov = kst['cov']
    return 0.5 * float(np.sum(np.log(np.diag(Lc)))) + float(res @ res) / (2 * cov ** 2) + N / 2 * math.log(2 * math.pi * cov ** 2)


def kriging_predict_w(kst, xS, w0):
    """y_kriging / mse_kriging of Gaussian_Process_Regression.ipynb:205-226, :236-253, 'gaussian' branch: the correlation
    vector is r = exp(-d^2 / (2 w0)) -- the variance-like parameterisation, where y_kriging_v2 has exp(-d^2 / (2 theta^2)) --
    against the K that gpr_poly built with its own theta (kst).  Returns (mu, mse) for a batch of points."""
    xS = np.asarray(xS, dtype=np.float64)
    r = kernel_matrix(kst['xT'], xS, np.full(xS.shape[1], math.sqrt(float(w0))), RBF)
    f = polymat(xS, kst['deg'])
    mu = f @ kst['beta'] + r.T @ kst['Kir']
    mse = kst['cov'] ** 2 * (1.0 - np.sum(r * np.linalg.solve(kst['K'], r), axis=0))
    return mu, mse


def gpr_descent(fun, dfun, alpha, beta, max_its, w0):
    """conj_gradient_descent of Gaussian_Process_Regression.ipynb:45-100 with norm = 'component' and the constant step
    (the settings of the notebook's cells, :479-486, :625-632): max_its - 1 steps of
        d = (1 - beta) d_prev + beta sign(grad) sqrt(len(w));  w = w - alpha d,
    no stopping test; returns (weight_history, cost_history), max_its entries each.  dfun stands in for autograd's grad(g)."""
    w = np.array(w0, dtype=np.float64, copy=True)
    wh, ch = [w.copy()], [fun(w)]
    dtemp = 0.0
    for k in range(1, int(max_its)):
        g = np.asarray(dfun(w), dtype=np.float64)
        d = (1 - beta) * dtemp + beta * np.sign(g) * math.sqrt(len(w))
        w = w - alpha * d
        wh.append(w.copy())
        ch.append(fun(w))
        dtemp = d
    return wh, ch


# --------------------------------------------------------------------------- synthetic inputs
_M64 = (1 << 64) - 1


def splitmix64_uniform(seed, count, offset=0):
    """Counter-based splitmix64 -> (u >> 11) * 2^-53 in [0,1).  Element i depends
    only on (seed, offset+i), so shards can generate their own slice.  The product
    package carries an identical generator (host/synth.py); the tests compare them."""
    i = np.arange(offset, offset + count, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def cg_optim(X, y, sigmaY, x0, alpha, beta, niter, normalize, kernel=0):
    """GP/BayesOpt_v3.m:457-484, one start, with mle = mlefcn and jmle = mle_grad (the restatements of mlefun :367-390 and
    jacmlefun :392-417).  Histories are allocated with niter rows and keep their zeros behind an early stop, as in MATLAB."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
    d = x0.shape[0]
    x_hist = np.zeros((niter, d)); x_hist[0] = x0                              # :460
    f_hist = np.zeros(niter); f_hist[0] = mlefcn(X, y, x0, sigmaY, kernel)     # :461
    jf_hist = np.zeros((niter, d)); jf_hist[0] = mle_grad(X, y, x0, sigmaY, kernel=kernel)   # :462
    dtemp = jf_hist[0].copy()                                                  # :463
    k, exitflag = 1, 0
    while exitflag == 0 and k < niter:                                         # :466
        jf_hist[k] = mle_grad(X, y, x_hist[k - 1], sigmaY, kernel=kernel)      # :467
        if normalize == 1:
            dd = (1 - beta) * dtemp + beta * jf_hist[k] / np.linalg.norm(jf_hist[k])          # :469
        elif normalize == 2:
            dd = (1 - beta) * dtemp + beta * np.sign(jf_hist[k]) * np.sqrt(d)                 # :471
        else:
            dd = (1 - beta) * dtemp + beta * jf_hist[k]                                       # :473
        x_hist[k] = x_hist[k - 1] - alpha * dd                                 # :475
        f_hist[k] = mlefcn(X, y, x_hist[k], sigmaY, kernel)                    # :476
        dtemp = dd                                                             # :477
        if np.linalg.norm(x_hist[k] - x_hist[k - 1]) <= 1e-9:                  # :478
            exitflag = 1
        k += 1
    return x_hist, f_hist, jf_hist, exitflag


def cg_optim_wrap(X, y, sigmaY, x0, starts, alpha, beta, niter, normalize, kernel=0, mm=50):
    """GP/BayesOpt_v3.m:486-514 with the random restarts given (starts (nruns - 1, d))."""
    xh, fh, _, _ = cg_optim(X, y, sigmaY, x0, alpha, beta, niter, normalize, kernel)
    fopt, xopt = np.mean(fh[niter - mm - 1:]), xh[-1]                          # :492-493
    for s0 in np.asarray(starts, dtype=np.float64).reshape(-1, len(xopt)):
        xh, fh, _, _ = cg_optim(X, y, sigmaY, s0, alpha, beta, niter, normalize, kernel)
        f0 = np.mean(fh[niter - mm - 1:])                                      # :506
        if fopt > f0:                                                          # :508
            xopt, fopt = xh[-1], f0
    return xopt.copy(), float(fopt)


def fminsearch(fun, x0, tol_x=1e-4, tol_fun=1e-4, max_iter=None, max_fun_evals=None):
    """Nelder-Mead simplex search with fminsearch's published constants and rules (Lagarias, Reeds, Wright & Wright,
    SIAM J. Optim. 9 (1998); what GP/BayesOpt_v2b.m:44-46 and GP/BayesOptFcn_v1.m:55-57 call), one start, sequential.
    Returns (x, f(x), exitflag, iterations, evaluations)."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
    n = x0.shape[0]
    max_iter = 200 * n if max_iter is None else max_iter
    max_fun_evals = 200 * n if max_fun_evals is None else max_fun_evals
    rho, chi, psi, sigma = 1.0, 2.0, 0.5, 0.5
    v = np.tile(x0[:, None], (1, n + 1))
    for j in range(n):
        v[j, j + 1] = 1.05 * x0[j] if x0[j] != 0 else 0.00025
    fv = np.array([fun(v[:, j].copy()) for j in range(n + 1)])
    order = np.argsort(fv, kind='stable')
    fv, v = fv[order], v[:, order]
    itercount, func_evals = 1, n + 1
    while func_evals < max_fun_evals and itercount < max_iter:
        if np.max(np.abs(fv[0] - fv[1:])) <= max(tol_fun, 10 * np.spacing(abs(fv[0]))) and \
                np.max(np.abs(v[:, 1:] - v[:, :1])) <= max(tol_x, 10 * np.spacing(abs(np.max(v[:, 0])))):
            break
        xbar = np.sum(v[:, :n], axis=1) / n
        xr = (1 + rho) * xbar - rho * v[:, -1]
        fxr = fun(xr.copy())
        func_evals += 1
        shrink = False
        if fxr < fv[0]:
            xe = (1 + rho * chi) * xbar - rho * chi * v[:, -1]
            fxe = fun(xe.copy())
            func_evals += 1
            if fxe < fxr:
                v[:, -1], fv[-1] = xe, fxe
            else:
                v[:, -1], fv[-1] = xr, fxr
        elif fxr < fv[n - 1]:
            v[:, -1], fv[-1] = xr, fxr
        elif fxr < fv[-1]:
            xc = (1 + psi * rho) * xbar - psi * rho * v[:, -1]
            fxc = fun(xc.copy())
            func_evals += 1
            if fxc <= fxr:
                v[:, -1], fv[-1] = xc, fxc
            else:
                shrink = True
        else:
            xcc = (1 - psi) * xbar + psi * v[:, -1]
            fxcc = fun(xcc.copy())
            func_evals += 1
            if fxcc < fv[-1]:
                v[:, -1], fv[-1] = xcc, fxcc
            else:
                shrink = True
        if shrink:
            for j in range(1, n + 1):
                v[:, j] = v[:, 0] + sigma * (v[:, j] - v[:, 0])
                fv[j] = fun(v[:, j].copy())
            func_evals += n
        order = np.argsort(fv, kind='stable')
        fv, v = fv[order], v[:, order]
        itercount += 1
    exitflag = 0 if (func_evals >= max_fun_evals or itercount >= max_iter) else 1
    return v[:, 0].copy(), float(fv[0]), exitflag, itercount, func_evals
