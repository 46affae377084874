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
