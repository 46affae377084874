#!/usr/bin/env python3
"""Extended-precision "truth" for the badly conditioned golden cases (sigmaY = 0: jitter 1e-15 n, cond up to 4e11).

TEST INFRASTRUCTURE, build container only (minutes of CPU).  For the inputs of a golden fixture (xTrain, yTrain, theta,
sigmaY as IEEE doubles, taken as exact) it evaluates the reference's formulas (BayesOpt/BayesOpt_v4.ipynb:119-124, 138-151,
164) in 50-digit decimal arithmetic -- kernel entries, both Cholesky factorisations, the triangular solves -- and stores
theta1, -logL, alpha and, for a strided subset of the candidates, the posterior mean and variance.  Against these numbers
the tests measure how far the REFERENCE's double-precision answer is from the mathematically exact one, and require the GPU
path to be no further away than a small multiple of that.

    python oracle/make_truth_fixtures.py [--out tests/golden] [--cases g3_d4_n512_s0,g5_dup_train_s0]
    python oracle/make_truth_fixtures.py --grid      # -logL on the sigmaY = 0 likelihood grid of fixture g6 (mlefcn, :154-165)
"""
import argparse
import os
import time
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 50
D = Decimal


def dec(a):
    return [D(float(v)) for v in np.asarray(a, dtype=np.float64).reshape(-1)]      # a double converts to Decimal exactly


def kernel_row(xi, Xrows, inv_th2):
    """exp(-1/2 sum_t (x_it - x_jt)^2 / theta_t^2) for every row j (BayesOpt_v4.ipynb:119-123)."""
    out = []
    for xj in Xrows:
        s = D(0)
        for a, b, w in zip(xi, xj, inv_th2):
            df = a - b
            s += df * df * w
        out.append((-s / 2).exp())
    return out


def cholesky(A):
    """Lower Cholesky factor of the symmetric matrix A (list of rows), row-oriented (Cholesky-Banachiewicz)."""
    n = len(A)
    L = [[D(0)] * n for _ in range(n)]
    for i in range(n):
        Li = L[i]
        for j in range(i + 1):
            Lj = L[j]
            s = A[i][j]
            for k in range(j):
                s -= Li[k] * Lj[k]
            if i == j:
                if s <= 0:
                    raise ArithmeticError('not positive definite at row %d' % i)
                Li[j] = s.sqrt()
            else:
                Li[j] = s / Lj[j]
    return L


def solve_lower(L, b):
    n = len(L)
    x = [D(0)] * n
    for i in range(n):
        s = b[i]
        Li = L[i]
        for k in range(i):
            s -= Li[k] * x[k]
        x[i] = s / Li[i]
    return x


def solve_upper_t(L, b):
    """Solve L^T x = b."""
    n = len(L)
    x = [D(0)] * n
    for i in range(n - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, n):
            s -= L[k][i] * x[k]
        x[i] = s / L[i][i]
    return x


def truth_case(g, stride):
    X = np.asarray(g['xTrain'], dtype=np.float64)
    y = dec(g['yTrain'])
    th = dec(g['theta'])
    sY = float(g['sigmaY'])
    n, d = X.shape
    eps = D('1e-15') * n if sY == 0 else D(sY) * D(sY)                      # :138-141
    inv_th2 = [1 / (t * t) for t in th]
    Xr = [dec(X[i]) for i in range(n)]
    t0 = time.time()
    K0 = [kernel_row(Xr[i], Xr, inv_th2) for i in range(n)]
    A = [[K0[i][j] + (eps if i == j else 0) for j in range(n)] for i in range(n)]
    L1 = cholesky(A)                                                          # :142
    a1 = solve_upper_t(L1, solve_lower(L1, y))                               # :143
    theta1 = sum(p * q for p, q in zip(y, a1)) / n                           # :144
    nll = D(n) / 2 * ((2 * D('3.14159265358979323846264338327950288419716939937510')).ln() + 1) + D(n) / 2 * theta1.ln() \
        + sum(L1[i][i].ln() for i in range(n))                               # :164
    A2 = [[theta1 * K0[i][j] + (eps if i == j else 0) for j in range(n)] for i in range(n)]
    L2 = cholesky(A2)                                                         # :146-147
    alpha = [v / theta1 for v in a1]                                         # :148
    print('  fit done in %.0f s' % (time.time() - t0), flush=True)
    Xs = np.asarray(g['xTest'], dtype=np.float64)
    idx = np.arange(0, Xs.shape[0], stride)
    mu, var = [], []
    for j in idx:
        ks = [theta1 * v for v in kernel_row(dec(Xs[j]), Xr, inv_th2)]       # :135, :146
        mu.append(sum(p * q for p, q in zip(ks, alpha)))                     # :149
        v = solve_lower(L2, ks)                                              # :150
        c = theta1 - sum(t * t for t in v)                                   # :151 (diagonal entry, Kss = theta1 * 1)
        var.append(c if c >= D('1e-60') else D(0))
    print('  %d candidates done in %.0f s' % (len(idx), time.time() - t0), flush=True)
    f = lambda v: np.array([float(t) for t in v])
    return dict(theta1=np.array(float(theta1)), nll=np.array(float(nll)), alpha=f(alpha), idx=idx, mu=f(mu), var=f(var),
                digits=np.array(getcontext().prec))


def truth_nll(X, y, th, sY):
    """-logL of mlefcn (BayesOpt_v4.ipynb:154-165) in extended precision."""
    n = X.shape[0]
    eps = D('1e-15') * n if sY == 0 else D(sY) * D(sY)
    inv_th2 = [1 / (t * t) for t in dec(th)]
    Xr = [dec(X[i]) for i in range(n)]
    yv = dec(y)
    A = [kernel_row(Xr[i], Xr, inv_th2) for i in range(n)]
    for i in range(n):
        A[i][i] += eps
    L1 = cholesky(A)
    a1 = solve_upper_t(L1, solve_lower(L1, yv))
    theta1 = sum(p * q for p, q in zip(yv, a1)) / n
    return D(n) / 2 * ((2 * D('3.14159265358979323846264338327950288419716939937510')).ln() + 1) + D(n) / 2 * theta1.ln() \
        + sum(L1[i][i].ln() for i in range(n))


def grid_truth(out):
    g = np.load(os.path.join(out, 'g6_mle_grid.npz'))
    X, y, th0 = np.asarray(g['xTrain'], dtype=np.float64), g['yTrain'], g['th0']
    t0 = time.time()
    T = np.empty((len(th0), len(th0)))
    for i, a in enumerate(th0):
        for j, b in enumerate(th0):
            T[i, j] = float(truth_nll(X, y, np.array([a, b]), 0.0))
        print('  row %d / %d  (%.0f s)' % (i + 1, len(th0), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(out, 'truth_g6_sigma0.npz'), nll=T, th0=th0, digits=np.array(getcontext().prec))
    err = np.abs(g['nll_sigma0'] - T) / np.abs(T)
    print('  reference vs truth on the grid: median %.2e  max %.2e' % (np.median(err), err.max()))


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument('--out', default=os.path.join(here, '..', 'tests', 'golden'))
    ap.add_argument('--cases', default='g3_d4_n512_s0,g5_dup_train_s0')
    ap.add_argument('--grid', action='store_true')
    a = ap.parse_args()
    if a.grid:
        grid_truth(a.out)
        return
    for name in a.cases.split(','):
        g = np.load(os.path.join(a.out, name + '.npz'))
        if int(g['raised_linalgerror']):
            print(name, ': the reference raised LinAlgError, no truth to compare with')
            continue
        print(name, flush=True)
        stride = 25 if len(g['xTrain']) > 100 else 3
        try:
            t = truth_case(g, stride)
        except ArithmeticError as e:
            print('  exact arithmetic says:', e)
            continue
        np.savez_compressed(os.path.join(a.out, 'truth_' + name + '.npz'), **t)
        rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
        print('  reference vs truth: nll %.2e  alpha %.2e  mu %.2e  var %.2e' % (
            abs(float(g['nll']) - float(t['nll'])) / abs(float(t['nll'])), rel(g['alpha'], t['alpha']),
            rel(g['Mu'][t['idx']], t['mu']), rel(g['var'][t['idx']], t['var'])), flush=True)


if __name__ == '__main__':
    main()
