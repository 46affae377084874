#!/usr/bin/env python3
"""How well is the reference's answer defined when K + eps I is badly conditioned?  (CPU only.)

Golden case g3_d4_n512_s0: sigmaY = 0 -> jitter 1e-15 n, cond 4e11.  theta1 = y^T (K + eps I)^-1 y / n scales every
posterior variance.  Prints how far theta1 and alpha move under changes that are all "the same algorithm in IEEE double":
  (a) Cholesky with another panel width (every operation LAPACK-quality: dpotrf on the panel, triangular solves, GEMM),
  (b) K's entries changed by one unit in the last place (what any other summation order of the distances does),
  (c) the reference's own value against an extended-precision evaluation.
The reference and the NumPy restatement agree to 1e-8 only because both call the same dpotrf on the same bits."""
import os
import sys

import numpy as np
from scipy.linalg import solve_triangular

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as o  # noqa: E402


def blocked_chol(A, pb):
    A = A.copy()
    n = A.shape[0]
    for p in range(0, n, pb):
        s = slice(p, min(n, p + pb))
        Lpp = np.linalg.cholesky(A[s, s])
        A[s, s] = Lpp
        if s.stop < n:
            r = slice(s.stop, n)
            A[r, s] = solve_triangular(Lpp, A[r, s].T, lower=True).T
            A[r, r] -= A[r, s] @ A[r, s].T
    return np.tril(A)


def theta1_alpha(L, y):
    a = solve_triangular(L, y, lower=True)
    a = solve_triangular(L, a, lower=True, trans='T')
    return float(y @ a) / len(y), a


def noise_floor(name='g3_d4_n512_s0', extended=True):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    X, y, th = g['xTrain'], g['yTrain'].reshape(-1), g['theta']
    n = len(y)
    st = o.fit(X, y, th, float(g['sigmaY']))
    K = o.kernel_matrix(X, X, th) + o.jitter(n, float(g['sigmaY'])) * np.eye(n)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    out = {'cond': float(g['condK']), 'panel': {}, 'ulp': []}
    for pb in (32, 64, 128, 256):
        t, a = theta1_alpha(blocked_chol(K, pb), y)
        out['panel'][pb] = (abs(t - st['theta1']) / st['theta1'], rel(a, st['alpha1']))
    rng = np.random.default_rng(0)
    for _ in range(3):
        E = K * (1 + (rng.integers(0, 2, K.shape) * 2 - 1) * 1.1e-16)
        E = np.tril(E) + np.tril(E, -1).T
        t, a = theta1_alpha(np.linalg.cholesky(E), y)
        out['ulp'].append((abs(t - st['theta1']) / st['theta1'], rel(a, st['alpha1'])))
    if extended:
        Kq, yq = K.astype(np.longdouble), y.astype(np.longdouble)
        Lq = np.zeros_like(Kq)
        for j in range(n):
            Lq[j, j] = np.sqrt(Kq[j, j] - Lq[j, :j] @ Lq[j, :j])
            Lq[j + 1:, j] = (Kq[j + 1:, j] - Lq[j + 1:, :j] @ Lq[j, :j]) / Lq[j, j]
        z = np.zeros(n, dtype=np.longdouble)
        for i in range(n):
            z[i] = (yq[i] - Lq[i, :i] @ z[:i]) / Lq[i, i]
        out['extended'] = abs(float(z @ z / n) - st['theta1']) / st['theta1']
    return out


if __name__ == '__main__':
    r = noise_floor()
    print('cond(K + eps I) = %.3g' % r['cond'])
    for pb, (t, a) in r['panel'].items():
        print('Cholesky with panel width %3d : theta1 moves by %.2e, alpha by %.2e (relative, max-norm)' % (pb, t, a))
    for t, a in r['ulp']:
        print('K entries +- 1 ulp            : theta1 moves by %.2e, alpha by %.2e' % (t, a))
    print('reference theta1 vs extended precision (x87 long double): %.2e' % r['extended'])
