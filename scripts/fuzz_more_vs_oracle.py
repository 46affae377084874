#!/usr/bin/env python3
"""Randomised differential run, second part: the likelihood sweep (-logL and analytic gradient through every internal path:
one workgroup per setting, lockstep batch, stream lanes, single setting), the frozen-theta append loop and the kriging
flavour, GPU library against the CPU oracle over random shapes (well-conditioned settings).
python3 scripts/fuzz_more_vs_oracle.py [cases per part] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP
from bayesian_optimization_amd import notebook_api as nb
from oracle import gp_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
gp = GP(0)
bad = 0
t0 = time.time()

# ---- part 1: -logL + gradient
worst = dict(nll=0.0, grad=0.0)
for c in range(cases):
    n = int(rng.choice([1, 2, 5, 33, 64, 65, 100, 128, 129, 200, 257, 400, 700, 1100]))
    d = int(rng.integers(1, 11))
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 70]))
    if n >= 700:
        B = min(B, 5)                      # the oracle side is O(B n^3)
    kernel = int(rng.integers(0, 2))
    X = rng.random((n, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n)
    th = rng.uniform(0.3, 0.9, (B, d)) * (1 + 0.15 * d)
    sg = rng.choice([0.05, 0.1, 0.3, 0.5], B)
    nll, g, st = gp.lml_batch(X, y, th, sg, kernel=kernel, grad=True)
    e = dict(nll=0.0, grad=0.0)
    for b in range(B):
        ref = o.mlefcn(X, y, th[b], float(sg[b]), kernel=kernel)
        gref = o.mle_grad(X, y, th[b], float(sg[b]), kernel=kernel)
        e['nll'] = max(e['nll'], abs(nll[b] - ref) / max(abs(ref), 1.0))
        e['grad'] = max(e['grad'], float(np.max(np.abs(g[b] - gref)) / max(np.max(np.abs(gref)), 1e-3 * n)))
    ok = int(np.all(st == 0)) and e['nll'] <= 1e-10 and e['grad'] <= 1e-8
    for k in worst:
        worst[k] = max(worst[k], e[k])
    if not ok:
        bad += 1
        print('MISMATCH lml', dict(n=n, d=d, B=B, kernel=kernel), e, st.tolist()[:8], flush=True)
print('lml: cases %d  mismatches so far %d  worst %s  (%.0f s)' % (cases, bad, {k: '%.2e' % v for k, v in worst.items()}, time.time() - t0), flush=True)

# ---- part 2: append loop, theta frozen (append == refit on the enlarged set)
worst = dict(mu=0.0, var=0.0, nll=0.0)
for c in range(cases):
    n0 = int(rng.choice([1, 3, 16, 100, 127, 128, 250, 383]))
    k = int(rng.choice([1, 2, 5, 20]))
    d = int(rng.integers(1, 7))
    m = int(rng.choice([1, 64, 257, 1000, 5000]))
    kernel = int(rng.integers(0, 2))
    sY = float(rng.choice([0.05, 0.1, 0.3]))
    theta = rng.uniform(0.3, 0.9, d) * (1 + 0.15 * d)
    X = rng.random((n0 + k, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n0 + k); Xs = rng.random((m, d))
    gp.reserve_appends(k)
    gp.fit(X[:n0], y[:n0], theta, sY, kernel=kernel, variant=bo.VARIANT_SINGLE)
    gp.set_candidates(Xs)
    gp.posterior()
    for i in range(n0, n0 + k):
        gp.append(X[i], y[i])
    mu, var = gp.read(mu=True, var=True)
    mu_o, var_o, st = o.gprfcn_diag(Xs, X, y, theta, sY, kernel=kernel, variant=bo.VARIANT_SINGLE)
    nll_g, n_now = gp.nll, gp.n
    e = dict(mu=rel(mu, mu_o), var=float(np.max(np.abs(var - var_o)) / max(np.max(np.abs(var_o)), 1e-300)),
             nll=abs(nll_g - st['nll']) / max(abs(st['nll']), 1.0))
    ok = e['mu'] <= 1e-9 and e['var'] <= 1e-8 and e['nll'] <= 1e-9 and int(n_now) == n0 + k
    for kk in worst:
        worst[kk] = max(worst[kk], e[kk])
    if not ok:
        bad += 1
        print('MISMATCH append', dict(n0=n0, k=k, d=d, m=m, kernel=kernel, sY=sY), e, flush=True)
gp.reserve_appends(0)
print('append: cases %d  mismatches so far %d  worst %s  (%.0f s)' % (cases, bad, {k: '%.2e' % v for k, v in worst.items()}, time.time() - t0), flush=True)

# ---- part 3: kriging with a polynomial trend (no jitter: length-scales short enough for a well-conditioned K)
worst = dict(beta=0.0, cov=0.0, mu=0.0, mse=0.0)
ran = 0
for c in range(cases):
    d = int(rng.integers(1, 4))
    deg = int(rng.integers(0, 4))
    n = int(rng.choice([12, 20, 40, 90, 150]))
    m = int(rng.choice([1, 7, 300, 2000]))
    xT = rng.random((n, d)) * 4.0
    yT = np.sin(xT.sum(1)) + 0.3 * xT[:, 0] + 0.05 * rng.standard_normal(n)
    xS = rng.random((m, d)) * 4.0
    theta = float(rng.uniform(0.08, 0.2)) * (1.0 if d == 1 else 2.0)
    kst = o.kriging_fit(xT, yT, theta, deg)
    if np.linalg.cond(kst['K']) > 1e6 or np.linalg.cond(kst['FT'].T @ np.linalg.solve(kst['K'], kst['FT'])) > 1e8:
        continue
    ran += 1
    mu_o, mse_o = o.kriging_predict(kst, xS)
    beta, cov, K, Ks, Kss, FT, FS = nb.gpr_poly(xT, yT, xT, theta, 'gaussian', deg)
    mu, mse = nb.y_kriging_v2(xS, beta, cov, K, FT, xT, yT, theta, deg, 'gaussian')
    e = dict(beta=rel(beta, kst['beta']), cov=abs(cov - kst['cov']) / abs(kst['cov']), mu=rel(mu, mu_o),
             mse=float(np.max(np.abs(mse - mse_o)) / max(np.max(np.abs(mse_o)), kst['cov'] ** 2)))
    ok = e['beta'] <= 1e-8 and e['cov'] <= 1e-8 and e['mu'] <= 1e-8 and e['mse'] <= 1e-8
    for kk in worst:
        worst[kk] = max(worst[kk], e[kk])
    if not ok:
        bad += 1
        print('MISMATCH kriging', dict(n=n, d=d, deg=deg, m=m, theta=theta), e, flush=True)
print('kriging: cases %d (of %d drawn; the rest ill-conditioned without jitter)  mismatches so far %d  worst %s  (%.0f s)' % (ran, cases, bad, {k: '%.2e' % v for k, v in worst.items()}, time.time() - t0), flush=True)

# ---- part 4: the descent of the hyper-parameter fit, all on the device (n <= 128) against the host loop over lml_batch
worst = dict(cost=0.0)
ran = 0
for c in range(max(cases // 4, 4)):
    n = int(rng.choice([1, 2, 7, 31, 32, 33, 50, 64, 65, 100, 127, 128]))
    d = int(rng.choice([1, 2, 3, 5, 8, 16]))
    nruns = int(rng.choice([1, 3, 10]))
    its = int(rng.choice([1, 15, 60]))
    norm = str(rng.choice(['component', 'full', 'none']))
    step = str(rng.choice(['normal', 'diminishing', 'backtrack']))
    sY = float(rng.choice([0.0, 0.02, 0.2]))
    X = rng.random((n, d)); y = np.sin(4 * X.sum(1)) + 0.05 * rng.standard_normal(n)
    x0 = rng.random((nruns, d)) + 0.05
    beta = float(rng.choice([0.7, 1.0]))
    res = []
    try:
        for dev in ('1', '0'):
            os.environ['GPBO_DESCENT_DEVICE'] = dev
            res.append(nb.hypopt(nruns, its, 0.01, beta, norm, step, X, y.reshape(-1, 1), None, sY, x0=x0.copy()))
    except (np.linalg.LinAlgError, FloatingPointError, ZeroDivisionError):
        continue                      # (a zero gradient under the 'full' norm, or no start with a finite cost: both raise, as the reference would)
    ran += 1
    (wa, ca, fa), (wb, cb, fb) = res
    same = list(fa) == list(fb)
    for r in range(nruns):
        same = same and len(wa[r]) == len(wb[r]) and np.array_equal(np.array(wa[r]), np.array(wb[r]), equal_nan=True)
        if same and len(ca[r]):
            fin = np.isfinite(np.array(cb[r]))
            same = same and np.array_equal(np.isfinite(np.array(ca[r])), fin)
            if fin.any():
                worst['cost'] = max(worst['cost'], float(np.max(np.abs(np.array(ca[r])[fin] - np.array(cb[r])[fin]) / np.maximum(np.abs(np.array(cb[r])[fin]), 1.0))))
    if not same or worst['cost'] > 1e-12:
        bad += 1
        print('MISMATCH descent', dict(n=n, d=d, nruns=nruns, its=its, norm=norm, step=step, sY=sY), flush=True)
os.environ.pop('GPBO_DESCENT_DEVICE', None)
print('descent: cases %d  mismatches so far %d  worst %s  (%.0f s)' % (ran, bad, {k: '%.2e' % v for k, v in worst.items()}, time.time() - t0), flush=True)

# ---- part 5: acquisition functions and the first-occurrence argmax on uploaded posteriors (zeros in the variance, exact ties)
worst = dict(ei=0.0, ucb=0.0, pi=0.0)
for c in range(cases):
    m = int(rng.choice([1, 2, 255, 256, 257, 2047, 2048, 2049, 4096, 10000, 65537, 300001]))
    mu = rng.standard_normal(m)
    var = rng.random(m) ** 3
    var[rng.random(m) < 0.05] = 0.0
    kappa = float(rng.choice([0.0, 0.01, 0.5]))
    dup = rng.integers(0, m, size=min(m, 5))
    gp.set_posterior(mu, var)
    for acq, name in ((bo.ACQ_EI, 'ei'), (bo.ACQ_UCB, 'ucb'), (bo.ACQ_PI, 'pi')):
        mode = int(rng.choice([bo.INCUMBENT_MAX_MU, bo.INCUMBENT_GIVEN]))
        inc = float(np.max(mu)) if mode == bo.INCUMBENT_MAX_MU else float(rng.standard_normal())
        if name == 'ei':
            ref = o.expected_improvement(mu.copy(), var, kappa, incumbent=inc)
        elif name == 'ucb':
            ref = o.ucb(mu, var, kappa)
        else:
            ref = o.probability_of_improvement(mu, inc, np.sqrt(var))
        # copies of the best candidate at random places: the first of the equal maxima must win
        ref = np.atleast_1d(ref)
        best = int(np.nanargmax(ref)) if np.isfinite(ref).any() else 0
        mu2, var2 = mu.copy(), var.copy()
        mu2[dup], var2[dup] = mu[best], var[best]
        if mode == bo.INCUMBENT_MAX_MU:
            inc = float(np.max(mu2))
        gp.set_posterior(mu2, var2)
        idx, top = gp.acquire(acq, kappa, mode, inc)
        vals = gp.read(mu=False, var=False, acq=True)
        if name == 'ei':
            ref = o.expected_improvement(mu2.copy(), var2, kappa, incumbent=inc)
        elif name == 'ucb':
            ref = o.ucb(mu2, var2, kappa)
        else:
            ref = o.probability_of_improvement(mu2, inc, np.sqrt(var2))
        ref = np.atleast_1d(ref)
        fin = np.isfinite(ref)
        diff = np.abs(vals[fin] - ref[fin])
        if name == 'ei':
            # deep in the lower tail imp Phi(Z) and sigma phi(Z) cancel: what either side can deliver is bounded by the
            # rounding of the two terms, not of their difference
            sg2 = np.sqrt(var2[fin])
            with np.errstate(divide='ignore', invalid='ignore'):
                terms = np.nan_to_num(sg2 * np.exp(-0.5 * ((mu2[fin] - inc - kappa) / sg2) ** 2), nan=0.0, posinf=0.0)
            diff = np.maximum(diff - 1e-14 * terms, 0.0)
        e = float(np.max(diff)) / max(float(np.max(np.abs(ref[fin]))), 1e-300) if fin.any() else 0.0
        if e > worst[name] and e > 1e-13 and fin.any():
            jw = int(np.argmax(diff))
            print('  (largest deviation so far: %s m=%d mode=%d kappa=%g inc=%.17g mu=%.17g var=%.17g got=%.17g ref=%.17g max|ref|=%.3g)'
                  % (name, m, mode, kappa, inc, mu2[fin][jw], var2[fin][jw], vals[fin][jw], ref[fin][jw], float(np.max(np.abs(ref[fin])))), flush=True)
        worst[name] = max(worst[name], e)
        nan_same = np.array_equal(np.isnan(vals), np.isnan(ref))
        own = int(np.argmax(vals))                      # np.argmax: the first NaN if there is one (0/0 where var = 0 and mu = incumbent)
        ok = e <= 1e-12 and nan_same and idx == own and (top == vals[own] or (np.isnan(top) and np.isnan(vals[own])))
        if not ok:
            bad += 1
            print('MISMATCH acquisition', dict(m=m, acq=name, mode=mode, kappa=kappa), e, idx, own, flush=True)
print('acquisition: cases %d x 3  mismatches so far %d  worst %s  (%.0f s)' % (cases, bad, {k: '%.2e' % v for k, v in worst.items()}, time.time() - t0), flush=True)
print('total mismatches', bad)
sys.exit(1 if bad else 0)
