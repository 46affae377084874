"""Hyper-parameter fit on device-evaluated likelihoods: the step right before the hot path in every BO iteration
(BayesOpt/BayesOpt_v4.ipynb:644).

Same update rule as the reference's conj_gradient_descent (BayesOpt_v4.ipynb:31-87) and the multi-start wrapper
hypopt / hypopt_wrap (:167-186), with two changes: the gradient is the analytic one (jacmlefun, GP/BayesOpt_v3.m:392-417)
evaluated by gpbo_lml_batch instead of autograd's reverse pass, and all nruns starts advance together, one batched
-logL + gradient call per iteration (the starts are the batch axis of the likelihood sweep).

Step rules: 'diminishing', 'backtrack' (the reference's own variant of the sufficient-decrease test, restated literally:
:62-71; at most 200 shrinking rounds per iteration) and the constant step (any other string, as in the reference where 'normal' / 'standard' fall through to the else
branch).  A start whose K stops being positive definite, or whose next iterate is not a set of finite non-zero length
scales any more, is frozen with cost +inf (the reference would raise LinAlgError out of the whole multi-start loop).

cg_optim / cg_optim_wrap restate the MATLAB scripts' version of the same descent (GP/BayesOpt_v3.m:457-514), which already
uses the analytic gradient.
"""
from __future__ import annotations

import os

import numpy as np

from .gp import GP, make_gp

_ctx = {}


def _lml_gp(device=None):
    """device=None: every visible GPU (the starts of a multi-start fit are the batch axis of the likelihood sweep and split
    over the devices); an int: that device only."""
    if device not in _ctx:
        _ctx[device] = make_gp() if device is None else GP(device)
    return _ctx[device]


def conj_gradient_descent_batched(xTrain, yTrain, sigmaY, alpha, beta, norm, steprule, max_its, W0, device=None, kernel=0):
    """All rows of W0 (nruns, d) advance in lockstep.  Returns (weight_history, cost_history, exitflag):
    weight_history[r] is the list of iterates of start r, cost_history[r] the -logL at each of them."""
    gp = _lml_gp(device)
    W = np.array(W0, dtype=np.float64, copy=True)
    nruns, d = W.shape
    n = np.asarray(xTrain).shape[0]
    if n <= 128 and d <= 16 and hasattr(gp, 'descent_small') and os.environ.get('GPBO_DESCENT_DEVICE', '1') != '0':
        # small training sets: the whole descent of every start runs inside one kernel launch (same rule, same arithmetic)
        wh, ch, cnt, flg = gp.descent_small(xTrain, yTrain, W, sigmaY, alpha, beta,
                                            0 if norm == 'component' else (1 if norm == 'full' else 2),
                                            2 if steprule == 'backtrack' else (1 if steprule == 'diminishing' else 0),
                                            int(max_its), kernel=kernel)
        w_hist = [[wh[r, i].copy() for i in range(cnt[r])] for r in range(nruns)]
        c_hist = [[float(ch[r, i]) for i in range(cnt[r])] for r in range(nruns)]
        return w_hist, c_hist, flg.astype(int)
    if steprule == 'backtrack':
        return _backtrack_batched(xTrain, yTrain, sigmaY, alpha, beta, norm, max_its, W0, device, kernel)
    tol = 1e-10
    sig = np.full(nruns, float(sigmaY))
    nll, grad, stat = gp.lml_batch(xTrain, yTrain, W, sig, kernel=kernel, grad=True)
    cost = np.where(stat == 0, nll, np.inf)
    w_hist = [[W[r].copy()] for r in range(nruns)]
    c_hist = [[cost[r]] for r in range(nruns)]
    exitflag = np.zeros(nruns, dtype=int)
    active = stat == 0
    dtemp = np.zeros_like(W)
    alpha_r = np.full(nruns, float(alpha))
    k = 1
    while k <= max_its and active.any():
        if norm == 'component':
            dstep = (1 - beta) * dtemp + beta * np.sign(grad) * np.sqrt(d)
        elif norm == 'full':
            with np.errstate(invalid='ignore', divide='ignore'):
                gn = grad / np.linalg.norm(grad, axis=1, keepdims=True)      # a zero gradient has no direction: NaN, handled below
            dstep = (1 - beta) * dtemp + beta * gn
        else:
            dstep = (1 - beta) * dtemp + beta * grad
        Wn = W - alpha_r[:, None] * dstep
        if steprule == 'diminishing':
            alpha_r[:] = 1.0 / k
        Wn[~active] = W[~active]
        moved = np.linalg.norm(Wn - W, axis=1)
        # an iterate that is not a set of length scales any more (NaN from a zero gradient under the 'full' norm, an
        # overflow, an exact zero) cannot be evaluated: the start is frozen exactly like one whose K is not positive definite
        unusable = ~np.all(np.isfinite(Wn) & (Wn != 0.0), axis=1)
        nll, gnew, stat = gp.lml_batch(xTrain, yTrain, np.where(unusable[:, None], W, Wn), sig, kernel=kernel, grad=True)
        stat = np.where(unusable, 1, stat)
        for r in range(nruns):
            if not active[r]:
                continue
            if stat[r] != 0:                       # left the positive-definite region: freeze at the last good point
                active[r] = False
                c_hist[r].append(np.inf)
                w_hist[r].append(Wn[r].copy())
                continue
            W[r] = Wn[r]
            grad[r] = gnew[r]
            cost[r] = nll[r]
            dtemp[r] = dstep[r]
            w_hist[r].append(W[r].copy())
            c_hist[r].append(cost[r])
            if moved[r] < tol:
                exitflag[r] = 1
                active[r] = False
        k += 1
    return w_hist, c_hist, exitflag


def _step_direction(norm, grad, dtemp, beta, d):
    if norm == 'component':
        return (1 - beta) * dtemp + beta * np.sign(grad) * np.sqrt(d)
    if norm == 'full':
        with np.errstate(invalid='ignore', divide='ignore'):
            gn = grad / np.linalg.norm(grad, axis=1, keepdims=True)      # normalised first, then weighted, as the reference (:53-54)
        return (1 - beta) * dtemp + beta * gn
    return (1 - beta) * dtemp + beta * grad


def _backtrack_batched(xTrain, yTrain, sigmaY, alpha, beta, norm, max_its, W0, device, kernel, max_inner=200):
    """steprule == 'backtrack' (BayesOpt_v4.ipynb:62-71), every start in lockstep.  Per outer iteration and start:
        wtemp = w ; w = w - alpha d ; h = g(wtemp) + grad(wtemp).(w - wtemp) + |w - wtemp|^2 / (2 alpha)
        while g(w) > h:  alpha *= 0.8 ; w = w - alpha d ; h = (same expression, old wtemp, new alpha) ; wtemp = w
        w = w - alpha d                      (the reference steps once more after the loop; alpha stays shrunk)
    Each round of the inner loop is one batched -logL + gradient call over the starts still in it."""
    gp = _lml_gp(device)
    W = np.array(W0, dtype=np.float64, copy=True)
    nruns, d = W.shape
    tol = 1e-10
    sig = np.full(nruns, float(sigmaY))

    def evaluate(P, rows):
        nll = np.full(nruns, np.inf)
        g = np.zeros((nruns, d))
        st = np.ones(nruns, dtype=np.int32)
        # a trial point that is not a set of finite non-zero length scales cannot be evaluated: it counts as a failed one
        rows = np.asarray(rows, dtype=int)
        rows = rows[np.all(np.isfinite(P[rows]) & (P[rows] != 0.0), axis=1)] if len(rows) else rows
        if len(rows):
            a, b, c = gp.lml_batch(xTrain, yTrain, P[rows], sig[rows], kernel=kernel, grad=True)
            nll[rows] = np.where(c == 0, a, np.inf)
            g[rows] = np.where((c == 0)[:, None], b, 0.0)
            st[rows] = c
        return nll, g, st

    cost, grad, stat = evaluate(W, np.arange(nruns))
    w_hist = [[W[r].copy()] for r in range(nruns)]
    c_hist = [[cost[r]] for r in range(nruns)]
    exitflag = np.zeros(nruns, dtype=int)
    active = stat == 0
    dtemp = np.zeros_like(W)
    alpha_r = np.full(nruns, float(alpha))
    k = 1
    while k <= max_its and active.any():
        rows = np.flatnonzero(active)
        dstep = _step_direction(norm, np.where(active[:, None], grad, 1.0), dtemp, beta, d)
        Wt, ft, gt = W.copy(), cost.copy(), grad.copy()
        Wn = W - alpha_r[:, None] * dstep
        fn, gn, sn = evaluate(Wn, rows)
        with np.errstate(invalid='ignore', over='ignore'):           # rows that just failed carry inf / NaN and are not looked at
            h = ft + np.sum(gt * (Wn - Wt), axis=1) + np.sum((Wn - Wt) ** 2, axis=1) / (2 * alpha_r)
        inner = active & (sn == 0) & (fn > h)
        failed = active & (sn != 0)
        rounds = 0
        while inner.any() and rounds < max_inner:
            ir = np.flatnonzero(inner)
            alpha_r[ir] *= 0.8
            Wn[ir] = Wn[ir] - alpha_r[ir, None] * dstep[ir]
            h[ir] = ft[ir] + np.sum(gt[ir] * (Wn[ir] - Wt[ir]), axis=1) + np.sum((Wn[ir] - Wt[ir]) ** 2, axis=1) / (2 * alpha_r[ir])
            f2, g2, s2 = evaluate(Wn, ir)
            Wt[ir], ft[ir], gt[ir] = Wn[ir], f2[ir], g2[ir]
            fn[ir] = f2[ir]
            failed |= inner & (s2 != 0)
            inner = inner & (s2 == 0) & (fn > h)
            rounds += 1
        Wn = Wn - alpha_r[:, None] * dstep
        good = active & ~failed
        f3, g3, s3 = evaluate(Wn, np.flatnonzero(good))
        failed |= good & (s3 != 0)
        moved_all = np.linalg.norm(Wn - W, axis=1)
        for r in rows:
            if failed[r]:                          # left the positive-definite region: freeze at the last good point
                active[r] = False
                c_hist[r].append(np.inf)
                w_hist[r].append(Wn[r].copy())
                continue
            moved = moved_all[r]
            W[r], cost[r], grad[r], dtemp[r] = Wn[r], f3[r], g3[r], dstep[r]
            w_hist[r].append(W[r].copy())
            c_hist[r].append(cost[r])
            if moved < tol:
                exitflag[r] = 1
                active[r] = False
        k += 1
    return w_hist, c_hist, exitflag


def hypopt(nruns, max_its, alpha, beta, normrule, steprule, xTrain, yTrain, Daa, sigmaY, x0=None, kernel=0):
    """Signature of BayesOpt_v4.ipynb:167-178 (Daa is accepted and ignored; distances are rebuilt on the device).
    x0: optional (nruns, d) starting points; default np.random.rand like the reference.  kernel: 0 RBF (the reference's),
    1 Matern-5/2."""
    d = np.asarray(xTrain).shape[1]
    if x0 is None:
        x0 = np.array([np.random.rand(d) for _ in range(int(nruns))])
    w_hist, c_hist, flags = conj_gradient_descent_batched(xTrain, yTrain, sigmaY, alpha, beta, normrule, steprule,
                                                          int(max_its), x0, kernel=kernel)
    return w_hist, c_hist, list(flags)


def hypopt_wrap(nruns, max_its, alpha, beta, normrule, steprule, xTrain, yTrain, Daa, sigmaY, x0=None, kernel=0):
    """Signature of BayesOpt_v4.ipynb:180-186: the final iterate of the start with the lowest final cost."""
    param_cont, cost_cont, flag_cont = hypopt(nruns, max_its, alpha, beta, normrule, steprule, xTrain, yTrain, Daa, sigmaY, x0, kernel)
    ff = np.array([c[-1] if np.isfinite(c[-1]) else (c[-2] if len(c) > 1 else np.inf) for c in cost_cont])
    if not np.isfinite(ff).any():
        # every start sits where K is not positive definite: np.linalg.cholesky raises here in the reference (:161)
        raise np.linalg.LinAlgError('hypopt_wrap: no start has a positive definite covariance matrix')
    iopt = int(np.argmin(ff))
    last = param_cont[iopt][-1] if np.isfinite(cost_cont[iopt][-1]) else param_cont[iopt][-2]
    return last


def cg_optim(xTrain, yTrain, sigmaY, x0, alpha, beta, niter, normalize, device=None, kernel=0):
    """The MATLAB scripts' descent on -logL (cg_optim, GP/BayesOpt_v3.m:457-484) with mle = mlefun (:367-390) and jmle =
    jacmlefun (:392-417) -- the analytic gradient the library evaluates -- for every row of x0 (R, d) in lockstep: one
    batched -logL + gradient call per iteration.  It differs from the notebook's conj_gradient_descent in four places,
    all kept: the momentum term starts from the raw gradient at x0 (dtemp = jf_hist(1,:)), 'component' is normalize == 2
    and 'full' normalize == 1, the stopping test is norm(step) <= 1e-9, and the histories are arrays of niter rows
    allocated up front, so a start that stops early leaves ZERO rows behind its last iterate (x_hist(end,:) is then zero).
    Returns (x_hist (R, niter, d), f_hist (R, niter), jf_hist (R, niter, d), exitflag (R,)).  MATLAB's chol raises where K
    is not positive definite; so does this (numpy.linalg.LinAlgError, naming the start)."""
    gp = _lml_gp(device)
    X0 = np.atleast_2d(np.array(x0, dtype=np.float64))
    R, d = X0.shape
    niter = int(niter)
    sig = np.full(R, float(sigmaY))
    x_hist = np.zeros((R, niter, d))
    f_hist = np.zeros((R, niter))
    jf_hist = np.zeros((R, niter, d))
    exitflag = np.zeros(R, dtype=int)

    def evaluate(P, rows):
        bad = rows[~np.all(np.isfinite(P[rows]) & (P[rows] != 0.0), axis=1)]
        if len(bad):
            raise np.linalg.LinAlgError('cg_optim: start %d left the set of finite non-zero length scales' % int(bad[0]))
        nll, g, st = gp.lml_batch(xTrain, yTrain, P[rows], sig[rows], kernel=kernel, grad=True)
        if np.any(st != 0):
            raise np.linalg.LinAlgError('cg_optim: start %d: matrix is not positive definite' % int(rows[np.flatnonzero(st != 0)[0]]))
        return nll, g

    rows = np.arange(R)
    x = X0.copy()
    f, g = evaluate(x, rows)
    x_hist[:, 0], f_hist[:, 0], jf_hist[:, 0] = x, f, g
    dtemp = g.copy()
    k = 1                                            # 0-based row being filled (MATLAB's k = 2)
    while k < niter and len(rows):
        jf_hist[rows, k] = g[rows]                   # jmle(x_hist(k-1,:)): the gradient of the previous evaluation
        gr = g[rows]
        if normalize == 1:
            dd = (1 - beta) * dtemp[rows] + beta * gr / np.linalg.norm(gr, axis=1, keepdims=True)
        elif normalize == 2:
            dd = (1 - beta) * dtemp[rows] + beta * np.sign(gr) * np.sqrt(d)
        else:
            dd = (1 - beta) * dtemp[rows] + beta * gr
        xn = x.copy()
        xn[rows] = x[rows] - alpha * dd
        fn, gn = evaluate(xn, rows)
        x_hist[rows, k] = xn[rows]
        f_hist[rows, k] = fn
        dtemp[rows] = dd
        stop = np.linalg.norm(xn[rows] - x[rows], axis=1) <= 1e-9
        g[rows] = gn
        x = xn
        exitflag[rows[stop]] = 1
        rows = rows[~stop]
        k += 1
    return x_hist, f_hist, jf_hist, exitflag


def cg_optim_wrap(xTrain, yTrain, sigmaY, x0, lb, ub, alpha, beta, nruns, niter, normalize, starts=None, rng=None, device=None,
                  kernel=0, mm=50):
    """cg_optim_wrap (GP/BayesOpt_v3.m:486-514): nruns descents, the first from x0, the others from uniform random points
    (in [lb, ub] when given, else in [0, 1)); the winner is the run with the lowest MEAN of its last mm + 1 = 51 costs
    (f1_hist(niter-mm:end)), and what is returned is that run's LAST history row -- zero if it stopped early, as in MATLAB.
    starts: optional (nruns - 1, d) array in place of the random draws (MATLAB's generator cannot be reproduced); rng: a
    numpy Generator for them otherwise.  All runs advance together.  Returns (xopt (d,), fopt)."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
    d = x0.shape[0]
    nruns, niter = int(nruns), int(niter)
    if niter <= mm:
        raise ValueError('cg_optim_wrap: niter must exceed mm = %d (f1_hist(niter-mm:end) starts at row niter - mm >= 1)' % mm)
    if starts is None:
        rng = np.random.default_rng() if rng is None else rng
        u = rng.random((max(nruns - 1, 0), d))
        no_box = (lb is None or np.size(lb) == 0) and (ub is None or np.size(ub) == 0)
        starts = u if no_box else (np.asarray(ub, dtype=np.float64) - np.asarray(lb, dtype=np.float64)) * u + np.asarray(lb, dtype=np.float64)
    starts = np.asarray(starts, dtype=np.float64).reshape(max(nruns - 1, 0), d)
    x_hist, f_hist, _, _ = cg_optim(xTrain, yTrain, sigmaY, np.vstack([x0[None, :], starts]), alpha, beta, niter, normalize,
                                    device=device, kernel=kernel)
    score = f_hist[:, niter - mm - 1:].mean(axis=1)
    best, fopt = 0, score[0]
    for i in range(1, nruns):                         # "if fopt > f0": a later run wins only if strictly better
        if fopt > score[i]:
            best, fopt = i, score[i]
    return x_hist[best, -1].copy(), float(fopt)


def gpr_descent(xT, yT, deg, epsilon, alpha, beta, max_its, W0, fd_h=1e-6, device=None):
    """The GPR notebook's hyper-parameter fit (Gaussian_Process_Regression.ipynb:473-510, :621-660): its own
    conj_gradient_descent (:45-100; no stopping test, max_its - 1 steps, histories of max_its entries) on gpr_mle (:255-263)
    with norm = 'component' and the constant step, every start of W0 (nruns,) in lockstep.  The notebook differentiates
    gpr_mle with autograd; under the 'component' rule only the SIGN of the derivative enters the step, and here it comes
    from a central difference of the device-evaluated gpr_mle, (f(w + h|w|) - f(w - h|w|)) / (2 h |w|): one library call per
    iteration evaluates the three points of every start.  Returns (weight_history, cost_history) as lists per start.  A
    start whose next point cannot be evaluated (K + epsilon I not positive definite) keeps its last point, cost +inf."""
    gp = _lml_gp(device)
    w = np.array(W0, dtype=np.float64, copy=True).reshape(-1)
    nruns = w.shape[0]

    def evaluate(pts):
        h = fd_h * np.abs(pts)
        vals, stat = gp.kriging_lml(xT, yT, np.concatenate([pts, pts + h, pts - h]), deg, epsilon)
        vals = np.where(stat == 0, vals, np.inf)
        f0, fp, fm = vals[:nruns], vals[nruns:2 * nruns], vals[2 * nruns:]
        with np.errstate(invalid='ignore'):
            g = (fp - fm) / (2 * h)
        return f0, np.where(np.isfinite(g), g, 0.0)

    cost, grad = evaluate(w)
    w_hist = [[np.array([w[r]])] for r in range(nruns)]
    c_hist = [[cost[r]] for r in range(nruns)]
    dtemp = np.zeros(nruns)
    for k in range(1, int(max_its)):
        d = (1 - beta) * dtemp + beta * np.sign(grad) * 1.0      # sqrt(len(w0)) = 1: theta is a scalar in this flavour
        wn = w - alpha * d
        ok = np.isfinite(wn) & (wn != 0.0)
        wn = np.where(ok, wn, w)
        cost, grad = evaluate(wn)
        w, dtemp = wn, d
        for r in range(nruns):
            w_hist[r].append(np.array([w[r]]))
            c_hist[r].append(cost[r])
    return w_hist, c_hist
