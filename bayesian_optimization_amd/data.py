"""The data side of the reference's measured-mixture experiment: the .mat file -> (x, y) objective, the candidate grid
and the pool of measured points a proposed candidate is snapped to.  Host-side plumbing (NumPy), no GPU work.

Reference: BayesOpt/BayesOpt_v4.ipynb:681-714 (Python), BayesOpt/GP/BayesOpt_v3.m:6-45 and :317-320 (MATLAB)."""
from __future__ import annotations

import numpy as np


def mixture_objective(IMbase, Cc, p, ff=1e5):
    """IMbase (N, P): one measured spectrum per mixture; Cc (2, N): the two concentrations of every mixture.
    Objective of mixture i = -mean((IMbase[i] - IMbase[p])**2), inputs = Cc[:, i] * ff
    (BayesOpt_v4.ipynb:687-697, BayesOpt_v3.m:12-26).  Returns x (N, 2), y (N, 1)."""
    IM = np.asarray(IMbase, dtype=np.float64)
    Cc = np.asarray(Cc, dtype=np.float64)
    if Cc.shape != (2, IM.shape[0]):
        raise ValueError('Cc must be (2, N) for IMbase (N, P); got %s and %s' % (Cc.shape, IM.shape))
    if not 0 <= p < IM.shape[0]:
        raise ValueError('target index p out of range')
    y = -np.mean((IM - IM[p]) ** 2, axis=1)[:, None]
    x = Cc.T * ff
    return np.ascontiguousarray(x), y


def load_mixture_mat(path, p=40, ff=1e5):
    """Reads fields IMbase and Cc of the processed .mat file (BayesOpt_v4.ipynb:682-685) -> mixture_objective."""
    from scipy.io import loadmat
    mat = loadmat(path)
    return mixture_objective(mat['IMbase'], mat['Cc'], p, ff)


def train_test_split(x, y, p, perm, nT=None):
    """Training set = the first nT entries of the permutation, target p removed (BayesOpt_v3.m:30-33; the notebook's
    np.delete at :700 discards its result, so there p may stay in the training set -- pass perm with p in it to get
    that behaviour).  Returns xT, yT, xTest (2,), yTest."""
    perm = np.asarray(perm)
    nT = len(x) - 1 if nT is None else nT
    ind = perm[:nT]
    return x[ind], y[ind], x[p], y[p]


def candidate_grid(x, xT, numpts=70):
    """Per axis: numpts equispaced values over the data range merged with the training abscissae, sorted and unique
    (np.unique); the candidates are their Cartesian product in lexicographic order, first axis slowest
    (BayesOpt_v4.ipynb:705-714).  Returns sx, sy, xp (len(sx) * len(sy), 2)."""
    sx = np.unique(np.hstack((np.linspace(np.min(x[:, 0]), np.max(x[:, 0]), numpts), xT[:, 0])))
    sy = np.unique(np.hstack((np.linspace(np.min(x[:, 1]), np.max(x[:, 1]), numpts), xT[:, 1])))
    xp = np.stack(np.meshgrid(sx, sy, indexing='ij'), -1).reshape(-1, 2)
    return sx, sy, xp


class MeasuredPool:
    """The measured points not yet in the training set.  take_nearest(x) removes and returns the one closest to x in
    the 1-norm, first one on ties: the experiment can only 'evaluate' mixtures that were measured
    (BayesOpt_v3.m:317-320)."""

    def __init__(self, xdat, ydat):
        self.x = np.array(xdat, dtype=np.float64, copy=True)
        self.y = np.array(ydat, dtype=np.float64, copy=True).reshape(len(self.x), -1)

    def __len__(self):
        return len(self.x)

    def take_nearest(self, x_query):
        if len(self.x) == 0:
            raise IndexError('the pool of measured points is empty')
        i2 = int(np.argmin(np.sum(np.abs(self.x - np.asarray(x_query, dtype=np.float64)), axis=1)))
        xs, ys = self.x[i2].copy(), self.y[i2].copy()
        self.x = np.delete(self.x, i2, axis=0)
        self.y = np.delete(self.y, i2, axis=0)
        return xs, ys
