"""Where does the sigmaY = 0 accuracy go?  Golden case g3_d4_n512_s0 (cond 4e11): every intermediate of the GPU fit against
the CPU oracle, and the posterior recomputed on the host from the GPU's own factor."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.linalg import solve_triangular
from oracle import gp_oracle as o
from bayesian_optimization_amd import GP
import bayesian_optimization_amd as bo
name = sys.argv[1] if len(sys.argv) > 1 else 'g3_d4_n512_s0'
g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
X, y, Xs, th, s = g['xTrain'], g['yTrain'].reshape(-1), g['xTest'], g['theta'], float(g['sigmaY'])
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
st = o.fit(X, y, th, s)
mu_o, var_o = o.predict(st, Xs)
print('oracle vs reference : mu %.2e var %.2e' % (rel(mu_o, g['Mu']), rel(var_o, g['var'])))
gp = GP(0)
for variant in (bo.VARIANT_V4, bo.VARIANT_SINGLE):
    gp.fit(X, y, th, s, variant=variant)
    mu, var, ei, idx, _, _ = gp.predict_ei(Xs, 0.01)
    Lg, ag = gp.factor()
    if variant == bo.VARIANT_V4:
        print('GPU vs reference    : mu %.2e var %.2e' % (rel(mu, g['Mu']), rel(var, g['var'])))
        print('GPU vs oracle       : mu %.2e var %.2e theta1 %.2e nll %.2e' % (rel(mu, mu_o), rel(var, var_o), abs(gp.theta1 - st['theta1']) / st['theta1'], abs(gp.nll - st['nll']) / abs(st['nll'])))
        print('factor L2 %.2e  alpha %.2e' % (rel(Lg, st['L2']), rel(ag, st['alpha'])))
        th1 = gp.theta1
        Ks = th1 * o.kernel_matrix(Xs, X, th)
        V = solve_triangular(Lg, Ks.T, lower=True)
        vh = th1 - np.sum(V * V, 0); vh[vh < 1e-60] = 0
        print('host posterior from the GPU factor + GPU theta1: var vs ref %.2e, vs GPU var %.2e' % (rel(vh, g['var']), rel(vh, var)))
        print('host mean from GPU alpha: vs ref %.2e vs GPU mu %.2e' % (rel(Ks @ ag, g['Mu']), rel(Ks @ ag, mu)))
        K2 = th1 * o.kernel_matrix(X, X, th) + st['eps'] * np.eye(len(y))
        print('residual |L L^T - M| / |M| GPU %.2e  numpy %.2e' % (rel(Lg @ Lg.T, K2), rel(np.linalg.cholesky(K2) @ np.linalg.cholesky(K2).T, K2)))
    else:
        so = o.fit(X, y, th, s, variant=o.VARIANT_SINGLE)
        m1, v1 = o.predict(so, Xs)
        print('single-factor GPU vs oracle: mu %.2e var %.2e L %.2e alpha %.2e' % (rel(mu, m1), rel(var, v1), rel(Lg, so['L1']), rel(ag, so['alpha1'])))
        V = solve_triangular(Lg, o.kernel_matrix(Xs, X, th).T, lower=True)
        vh = gp.theta1 * (1 - np.sum(V * V, 0)); vh[vh < 1e-60] = 0
        print('   host posterior from GPU factor: var vs oracle %.2e vs GPU %.2e' % (rel(vh, v1), rel(vh, var)))
