"""Continuous maximisation of the acquisition, the step right after the grid sweep in the MATLAB loops
(fminsearch on -acquisition_fcn from one random start: GP/BayesOpt_v2b.m:41-46, GP/BayesOptFcn_v1.m:52-57).

Here: K starts seeded from the best grid candidates, refined together by a compass (pattern) search; every iteration
evaluates all K*(2d+1) trial points in ONE posterior + acquisition call on the GPU (the trial points are the candidate
set of a second context that shares nothing with the grid sweep but the fitted model)."""
from __future__ import annotations

import numpy as np

from . import _lib as L


def top_k(values, k):
    """Indices of the k largest entries, best first, ties broken towards the lower index."""
    values = np.asarray(values)
    k = min(k, len(values))
    part = np.argpartition(-values, k - 1)[:k]
    return part[np.lexsort((part, -values[part]))]


def maximize_acquisition(gp, seeds, lo, hi, acq=L.ACQ_EI, kappa=0.01, incumbent=None, iters=40, step0=None, tol=1e-8):
    """gp: a fitted GP whose resident candidate set may be overwritten.  seeds (K, d) starting points; [lo, hi] box.
    incumbent: the fixed EI/PI incumbent (e.g. max(Mu) of the grid sweep); required for ACQ_EI / ACQ_PI.
    Returns (x_best (d,), value, all_x (K, d), all_values (K,))."""
    X = np.array(seeds, dtype=np.float64, copy=True)
    K, d = X.shape
    lo = np.broadcast_to(np.asarray(lo, dtype=np.float64), (d,))
    hi = np.broadcast_to(np.asarray(hi, dtype=np.float64), (d,))
    if acq in (L.ACQ_EI, L.ACQ_PI) and incumbent is None:
        raise ValueError('EI / PI need a fixed incumbent during the local search')
    mode = L.INCUMBENT_GIVEN if incumbent is not None else L.INCUMBENT_MAX_MU
    inc = 0.0 if incumbent is None else float(incumbent)
    step = np.full(K, float(step0) if step0 is not None else 0.05 * float(np.max(hi - lo)))
    dirs = np.vstack([np.zeros((1, d)), np.eye(d), -np.eye(d)])          # centre + 2d compass points

    def evaluate(P):
        gp.set_candidates(P.reshape(-1, d))
        gp.posterior()
        gp.acquire(acq, kappa, mode, inc)
        return gp.read(mu=False, var=False, acq=True).reshape(P.shape[0], P.shape[1])

    val = None
    for _ in range(iters):
        P = np.clip(X[:, None, :] + step[:, None, None] * dirs[None, :, :], lo, hi)      # (K, 2d+1, d)
        A = evaluate(P)
        best = np.argmax(A, axis=1)                                                     # first occurrence: centre wins ties
        val = A[np.arange(K), best]
        moved = best != 0
        X = P[np.arange(K), best]
        step = np.where(moved, step, 0.5 * step)
        if np.all(step < tol):
            break
    b = int(np.argmax(val))
    return X[b].copy(), float(val[b]), X, val


def fminsearch(gp, x0, acq=L.ACQ_UCB, kappa=0.01, incumbent=None, tol_x=1e-4, tol_fun=1e-4, max_iter=None, max_fun_evals=None):
    """The MATLAB loops' continuous step as they run it: fminsearch on -acquisition_fcn from a random start
    (GP/BayesOpt_v2b.m:41-46: A = @(x) -acquisition_fcn_ucb(x, kappa, ...), optimset('TolFun',1e-8,'TolX',1e-8);
    GP/BayesOptFcn_v1.m:52-57) -- the Nelder-Mead simplex method with fminsearch's published constants (Lagarias, Reeds,
    Wright & Wright 1998: reflection 1, expansion 2, contraction 1/2, shrink 1/2; initial simplex = x0 and x0 with one
    coordinate enlarged by 5 %, or set to 0.00025 where it is zero; stop when the simplex values lie within
    max(TolFun, 10 eps(f1)) AND the vertices within max(TolX, 10 eps(max|x1|)) of the best one, or after 200 d iterations
    or function evaluations), for every row of x0 (R, d) at once: each of the up to three evaluation rounds of an iteration
    (reflection; expansion or contraction; shrink) is ONE posterior + acquisition call over the trial points of all starts.
    Unconstrained, as in the reference.  MATLAB cannot run here: this follows the published method, parity with fminsearch's
    own arithmetic is unpinned.  gp: a fitted context whose resident candidates may be overwritten.
    Returns (x (R, d), acquisition value at x (R,), exitflag (R,) [1 converged, 0 limits reached], iterations (R,), evaluations (R,))."""
    X0 = np.atleast_2d(np.array(x0, dtype=np.float64))
    R, d = X0.shape
    if acq in (L.ACQ_EI, L.ACQ_PI) and incumbent is None:
        raise ValueError('EI / PI need a fixed incumbent during the search')
    mode = L.INCUMBENT_GIVEN if incumbent is not None else L.INCUMBENT_MAX_MU
    inc = 0.0 if incumbent is None else float(incumbent)
    max_iter = 200 * d if max_iter is None else int(max_iter)
    max_fun_evals = 200 * d if max_fun_evals is None else int(max_fun_evals)

    def fneg(P):
        """-acquisition at the rows of P (k, d), one library round trip."""
        if len(P) == 0:
            return np.empty(0)
        gp.set_candidates(P)
        gp.posterior()
        gp.acquire(acq, kappa, mode, inc)
        return -gp.read(mu=False, var=False, acq=True)

    V = np.repeat(X0[:, None, :], d + 1, axis=1)                       # (R, d+1, d): vertex 0 = x0
    for i in range(d):
        V[:, i + 1, i] = np.where(X0[:, i] != 0.0, 1.05 * X0[:, i], 0.00025)
    F = fneg(V.reshape(-1, d)).reshape(R, d + 1)
    order = np.argsort(F, axis=1, kind='stable')
    V = np.take_along_axis(V, order[:, :, None], axis=1)
    F = np.take_along_axis(F, order, axis=1)
    iters = np.ones(R, dtype=int)
    evals = np.full(R, d + 1, dtype=int)
    flag = np.zeros(R, dtype=int)
    active = np.ones(R, dtype=bool)
    eps = np.finfo(np.float64).eps
    while True:
        active &= (evals < max_fun_evals) & (iters < max_iter)
        small_f = np.max(np.abs(F[:, :1] - F[:, 1:]), axis=1) <= np.maximum(tol_fun, 10 * np.spacing(np.abs(F[:, 0])))
        small_x = np.max(np.abs(V[:, 1:, :] - V[:, :1, :]), axis=(1, 2)) <= np.maximum(tol_x, 10 * np.spacing(np.abs(np.max(V[:, 0, :], axis=1))))
        done = active & small_f & small_x
        flag[done] = 1
        active &= ~done
        rows = np.flatnonzero(active)
        if len(rows) == 0:
            break
        xbar = V[rows, :d, :].sum(axis=1) / d
        xlast = V[rows, d, :]
        xr = 2.0 * xbar - 1.0 * xlast                                     # (1 + rho) xbar - rho v_end
        fr = fneg(xr)
        evals[rows] += 1
        f1, fn, fl = F[rows, 0], F[rows, d - 1], F[rows, d]
        expand = fr < f1
        accept_r = ~expand & (fr < fn)
        outside = ~expand & ~accept_r & (fr < fl)
        inside = ~expand & ~accept_r & ~outside
        x2 = np.where(expand[:, None], 3.0 * xbar - 2.0 * xlast,           # (1 + rho chi) xbar - rho chi v_end
                      np.where(outside[:, None], 1.5 * xbar - 0.5 * xlast,  # (1 + psi rho) xbar - psi rho v_end
                               0.5 * xbar + 0.5 * xlast))                  # (1 - psi) xbar + psi v_end
        need2 = expand | outside | inside
        f2 = np.full(len(rows), np.inf)
        f2[need2] = fneg(x2[need2])
        evals[rows[need2]] += 1
        newx, newf = xr.copy(), fr.copy()
        take2 = (expand & (f2 < fr)) | (outside & (f2 <= fr)) | (inside & (f2 < fl))
        newx[take2], newf[take2] = x2[take2], f2[take2]
        shrink = (outside & ~(f2 <= fr)) | (inside & ~(f2 < fl))
        keep = ~shrink
        V[rows[keep], d, :] = newx[keep]
        F[rows[keep], d] = newf[keep]
        if shrink.any():
            rs = rows[shrink]
            V[rs, 1:, :] = V[rs, :1, :] + 0.5 * (V[rs, 1:, :] - V[rs, :1, :])
            F[rs, 1:] = fneg(V[rs, 1:, :].reshape(-1, d)).reshape(len(rs), d)
            evals[rs] += d
        order = np.argsort(F[rows], axis=1, kind='stable')
        V[rows] = np.take_along_axis(V[rows], order[:, :, None], axis=1)
        F[rows] = np.take_along_axis(F[rows], order, axis=1)
        iters[rows] += 1
    return V[:, 0, :].copy(), -F[:, 0], flag, iters, evals
