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
