"""Hyper-parameter fit on device-evaluated likelihoods: the step right before the hot path in every BO iteration
(BayesOpt/BayesOpt_v4.ipynb:644).

Same update rule as the reference's conj_gradient_descent (BayesOpt_v4.ipynb:31-87) and the multi-start wrapper
hypopt / hypopt_wrap (:167-186), with two changes: the gradient is the analytic one (jacmlefun, GP/BayesOpt_v3.m:392-417)
evaluated by gpbo_lml_batch instead of autograd's reverse pass, and all nruns starts advance together, one batched
-logL + gradient call per iteration (the starts are the batch axis of the likelihood sweep).

Supported step rules: 'diminishing' and the constant step (any other string, as in the reference where 'normal' /
'standard' fall through to the else branch).  'backtrack' is not implemented (data-dependent inner loops per start).
A start whose K stops being positive definite is frozen with cost +inf (the reference would raise LinAlgError out of
the whole multi-start loop).
"""
from __future__ import annotations

import numpy as np

from .gp import GP

_ctx = {}


def _lml_gp(device=0):
    if device not in _ctx:
        _ctx[device] = GP(device)
    return _ctx[device]


def conj_gradient_descent_batched(xTrain, yTrain, sigmaY, alpha, beta, norm, steprule, max_its, W0, device=0, kernel=0):
    """All rows of W0 (nruns, d) advance in lockstep.  Returns (weight_history, cost_history, exitflag):
    weight_history[r] is the list of iterates of start r, cost_history[r] the -logL at each of them."""
    if steprule == 'backtrack':
        raise NotImplementedError("steprule 'backtrack' is not available in the batched optimiser")
    gp = _lml_gp(device)
    W = np.array(W0, dtype=np.float64, copy=True)
    nruns, d = W.shape
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
            gn = grad / np.linalg.norm(grad, axis=1, keepdims=True)
            dstep = (1 - beta) * dtemp + beta * gn
        else:
            dstep = (1 - beta) * dtemp + beta * grad
        Wn = W - alpha_r[:, None] * dstep
        if steprule == 'diminishing':
            alpha_r[:] = 1.0 / k
        Wn[~active] = W[~active]
        moved = np.linalg.norm(Wn - W, axis=1)
        nll, gnew, stat = gp.lml_batch(xTrain, yTrain, Wn, sig, kernel=kernel, grad=True)
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
    iopt = int(np.argmin(ff))
    last = param_cont[iopt][-1] if np.isfinite(cost_cont[iopt][-1]) else param_cont[iopt][-2]
    return last
