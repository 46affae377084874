"""The reference's outer BO loop over a finite candidate set (BayesOpt/BayesOpt_v4.ipynb:618-664): per iteration refit the
length scales by multi-start descent on the likelihood (:644), posterior + EI over every candidate (:645-646), first-occurrence
argmax (:647), append the picked point (:655), stop once max EI < ei_tol (:663-664).

Every numeric step runs on the device: the likelihood sweep of the starts, the fit, the candidate sweep and the argmax.  The
candidates are uploaded once and stay resident across the refits; only the picked index, max EI and the fitted theta come
back per iteration (mu / diag(Cov) / ei too when keep_history is set, as the reference's *_hist lists do).
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .gp import make_gp
from .hyperopt import hypopt_wrap


def bayes_opt(x, y, xT, yT, sigmaY=0.0, kappa=0.01, ei_tol=1e-9, nruns=10, max_its=1e3, alpha=0.01, beta=1,
              normrule='component', steprule='normal', max_iter=None, kernel=L.KERNEL_RBF, keep_history=False,
              starts=None, callback=None, device=None, gp=None, devices=None):
    """x (m, d): the candidate set; y: its objective values (m,) / (m, 1), or a callable evaluated at the picked row.
    xT (n0, d), yT (n0, 1): the initial training set.  starts: optional callable (iteration, d) -> (nruns, d) starting points of
    the descent (default np.random.rand per start, as the reference).  callback(k, state) sees every iteration's record.
    devices: GPU ids to split the candidate set over (default: every visible GPU; ``device``: a single one).
    Returns a dict: xT, yT (grown), theta_hist, jj_hist, ei_max_hist, exitflag (1: EI fell below ei_tol, 0: max_iter reached)
    and, with keep_history, Mu_hist / Var_hist / EI_hist."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError('x must be (m, d)')
    xT = np.array(xT, dtype=np.float64, copy=True).reshape(-1, x.shape[1])
    yT = np.array(yT, dtype=np.float64, copy=True).reshape(-1, 1)
    if xT.shape[0] != yT.shape[0]:
        raise ValueError('xT has %d rows, yT %d' % (xT.shape[0], yT.shape[0]))
    y_of = y if callable(y) else None
    y_tab = None if callable(y) else np.asarray(y, dtype=np.float64).reshape(-1)
    if y_tab is not None and y_tab.shape[0] != x.shape[0]:
        raise ValueError('y has %d entries, x %d rows' % (y_tab.shape[0], x.shape[0]))
    gp = gp or make_gp(devices if devices is not None else ([device] if device is not None else None))
    out = {'theta_hist': [], 'jj_hist': [], 'ei_max_hist': [], 'exitflag': 0}
    if keep_history:
        out.update(Mu_hist=[], Var_hist=[], EI_hist=[])
    k = 1
    resident = False
    while max_iter is None or k <= max_iter:
        x0 = None if starts is None else np.asarray(starts(k, x.shape[1]), dtype=np.float64)
        theta = hypopt_wrap(int(nruns), max_its, alpha, beta, normrule, steprule, xT, yT, None, sigmaY, x0=x0, kernel=kernel)
        gp.fit(xT, yT, theta, sigmaY, kernel=kernel)
        if not resident:
            gp.set_candidates(x)
            resident = True
        gp.posterior()
        jj, ei_max = gp.acquire(L.ACQ_EI, kappa)
        x_next = x[jj]
        y_next = float(y_of(x_next)) if y_of is not None else float(y_tab[jj])
        out['theta_hist'].append(np.array(theta, copy=True))
        out['jj_hist'].append(jj)
        out['ei_max_hist'].append(ei_max)
        if keep_history:
            mu, var, ei = gp.read(mu=True, var=True, acq=True, count=x.shape[0])
            out['Mu_hist'].append(mu)
            out['Var_hist'].append(var)
            out['EI_hist'].append(ei)
        if callback is not None:
            callback(k, {'theta': theta, 'jj': jj, 'ei_max': ei_max, 'xT': xT, 'yT': yT, 'gp': gp})
        xT = np.vstack((xT, x_next))
        yT = np.vstack((yT, [[y_next]]))
        k += 1
        if ei_max < ei_tol:
            out['exitflag'] = 1
            break
    out['xT'], out['yT'] = xT, yT
    return out


def bayes_opt_fcn_v1(fun, kappa, x0=None, maxit=20, sigmaY=0.0, paropt=None, verbose=0, rng=None, gp=None, device=0):
    """The MATLAB scripts' BO loop as a function (GP/BayesOptFcn_v1.m:1-61), over a CONTINUOUS domain: per iteration the
    length scales are refitted by cg_optim_wrap on (mlefun, jacmlefun) from x0 = 0.05 + 2.95 rand (:46-50), the next point
    is fminsearch's minimiser of -acquisition_fcn(x) = -(Mu - kappa sqrt(diag(Cov))) from a uniform random start (:52-55;
    single-factor posterior, :68-84), and the loop runs maxit + 1 times or until K stops being positive definite (:29-40).
    fun: callable on a (d,) point.  paropt: dict with nruns, niter, alpha, beta, normalize, lb, ub (defaults :16-21).
    rng: numpy Generator for the three kinds of random draws (MATLAB's generator cannot be reproduced).
    Every numeric step runs on the device: the descent's likelihoods and gradients in lockstep over the nruns starts, the
    posterior and the acquisition under the simplex search.  Returns (xT (k, d), yT (k, 1), theta_hist)."""
    from . import acqopt
    from .gp import GP
    from .hyperopt import cg_optim_wrap
    rng = np.random.default_rng() if rng is None else rng
    x0 = rng.random(2) if x0 is None or np.size(x0) == 0 else np.asarray(x0, dtype=np.float64).reshape(-1)
    d = x0.shape[0]
    par = dict(nruns=200, niter=1000, alpha=0.01, beta=1, normalize=0, lb=None, ub=None)
    par.update(paropt or {})
    gp = gp or GP(device)
    xT = x0[None, :].copy()
    yT = np.array([[float(fun(x0))]])
    theta_hist = []
    i = 0
    while i <= maxit:
        if verbose:
            print('Iteration-%d; FunEval = %4.2f' % (i, yT[-1, 0]))
        th0 = (3 - 0.05) * rng.random(d) + 0.05
        starts = rng.random((par['nruns'] - 1, d))
        if par['lb'] is not None and np.size(par['lb']):
            starts = (np.asarray(par['ub'], dtype=np.float64) - np.asarray(par['lb'], dtype=np.float64)) * starts + np.asarray(par['lb'], dtype=np.float64)
        xstart = rng.random((1, d))
        try:
            theta, _ = cg_optim_wrap(xT, yT, sigmaY, th0, par['lb'], par['ub'], par['alpha'], par['beta'], par['nruns'], par['niter'],
                                     par['normalize'], starts=starts, device=device)
            theta_hist.append(np.array(theta, copy=True))
            if np.all(np.isfinite(theta) & (theta != 0.0)):
                gp.fit(xT, yT, theta, sigmaY, variant=L.VARIANT_SINGLE)
                # maximise Mu - kappa sigma: the library's upper-confidence form with the sign of kappa turned
                xs = acqopt.fminsearch(gp, xstart, L.ACQ_UCB, -float(kappa))[0][0]
            else:
                # the winning descent stopped early, so its last history row -- the returned theta -- is zero (with one training
                # point the gradient is zero and every run stops at once): the kernel is then 0 away from the training points,
                # the acquisition is flat, and fminsearch shrinks its simplex onto its starting vertex
                xs = xstart[0]
        except np.linalg.LinAlgError:
            return xT, yT, theta_hist                # the positive-definiteness stop (:29-40; MATLAB's chol inside the descent errors instead)
        xT = np.vstack([xT, xs[None, :]])
        yT = np.vstack([yT, [[float(fun(xs))]]])
        i += 1
    return xT, yT, theta_hist
