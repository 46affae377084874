#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  It json-loads the two
notebooks, exec()s their function-definition cells under plain NumPy (the
notebooks import autograd.numpy, which is absent; the hot-path functions only use
the NumPy API), calls the reference's own functions on seeded inputs and stores
inputs + outputs as small .npz fixtures.  Only derived numeric arrays are written;
no reference source text is copied.

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import json
import os
import warnings

import numpy as np
import scipy
import scipy.io


def load_function_cell(path, must_define):
    nb = json.load(open(path))
    code_cells = [''.join(c['source']) for c in nb['cells'] if c['cell_type'] == 'code']
    ns = {'np': np, 'sp': scipy, 'plt': None, 'gridspec': None}
    for src in code_cells:
        if all(('def ' + name + '(') in src for name in must_define):
            exec(compile(src, path, 'exec'), ns)
            return ns
    raise RuntimeError('function cell not found in ' + path)


def meta():
    return dict(numpy_version=np.__version__, scipy_version=scipy.__version__)


def run_gprfcn(ref, xs, xT, yT, theta, sigmaY, kappa, keep_factor):
    """Call the reference's gprfcn/expected_improvement/mlefcn; keep what callers consume."""
    out = dict(xTest=xs, xTrain=xT, yTrain=yT, theta=np.asarray(theta, float), sigmaY=float(sigmaY),
               kappa=float(kappa))
    try:
        Mu, Cov, K, Ks, Kss, L, alpha, v = ref['gprfcn'](xs, xT, yT, np.asarray(theta, float), sigmaY)
    except np.linalg.LinAlgError:
        out['raised_linalgerror'] = np.array(1)
        return out
    out['raised_linalgerror'] = np.array(0)
    ei = ref['expected_improvement'](Mu, Cov, kappa)
    Daa = ref['distmat_v2'](xT, xT)
    nll = ref['mlefcn'](xT, yT, Daa, np.asarray(theta, float), sigmaY)
    out.update(Mu=Mu.reshape(-1), var=np.diag(Cov).copy(), ei=ei, argmax=np.array(int(np.argmax(ei))),
               alpha=alpha.reshape(-1), nll=np.array(float(np.squeeze(nll))),
               condK=np.array(np.linalg.cond(K + 0.0)))
    if keep_factor:
        out['L'] = L
    return out


def save(outdir, name, d):
    d = dict(d)
    d.update({k: np.array(v) for k, v in meta().items()})
    np.savez_compressed(os.path.join(outdir, name + '.npz'), **d)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in d.items() if hasattr(v, 'shape') and v.size > 1})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    warnings.simplefilter('ignore')
    v4 = load_function_cell(os.path.join(a.ref, 'BayesOpt', 'BayesOpt_v4.ipynb'),
                            ['gprfcn', 'mlefcn', 'expected_improvement'])
    gpr = load_function_cell(os.path.join(a.ref, 'BayesOpt', 'Gaussian_Process_Regression.ipynb'),
                             ['gpr_poly', 'y_kriging_v2', 'polymat'])

    # ---- G1: the .mat mixture data, as the "2D Mixture Data" cells do (BayesOpt_v4.ipynb:681-807)
    mat = scipy.io.loadmat(os.path.join(a.ref, 'BayesOpt', 'GP', 'red3yellow3_4.15.19_processed.mat'))
    IMcc, Cc = mat['IMbase'], mat['Cc']
    p = 40
    X1 = IMcc[p, :]
    N = IMcc.shape[0]
    x = np.zeros((N, 2))
    y = np.zeros((N, 1))
    for i in range(N):
        y[i] = -np.mean((IMcc[i, :] - X1) ** 2)
        x[i, :] = Cc[:, i]
    x = x * 1e5
    nT = N - 1
    ind = np.random.default_rng(20190415).permutation(N)
    xT, yT = x[ind[:nT], :], y[ind[:nT]]
    numpts = 70
    sx = np.unique(np.hstack((np.linspace(np.min(x[:, 0]), np.max(x[:, 0]), numpts), xT[:, 0])))
    sy = np.unique(np.hstack((np.linspace(np.min(x[:, 1]), np.max(x[:, 1]), numpts), xT[:, 1])))
    xp = np.array([[sx[i], sy[j]] for i in range(len(sx)) for j in range(len(sy))])
    theta = np.array([0.1036165, 0.0995048])          # the author's saved fit, BayesOpt_v4.ipynb:776
    g1 = run_gprfcn(v4, xp, xT, yT, theta, 0.0, 0.01, keep_factor=True)
    g1.update(x_all=x, y_all=y, perm=ind, grid_sx=sx, grid_sy=sy)
    save(a.out, 'g1_mat_mixture', g1)

    # ---- G6: likelihood surface on G1's data (analogue of the 500x500 sweep at :727-731), 16x16
    th0 = np.linspace(0.05, 2.0, 16)
    Daa = v4['distmat_v2'](xT, xT)
    M = np.zeros((16, 16))
    for i in range(16):
        for j in range(16):
            M[i, j] = np.squeeze(v4['mlefcn'](xT, yT, Daa, np.array([th0[i], th0[j]]), 0.0))
    M2 = np.zeros((16, 16))
    for i in range(16):
        for j in range(16):
            M2[i, j] = np.squeeze(v4['mlefcn'](xT, yT, Daa, np.array([th0[i], th0[j]]), 0.05))
    # central differences of the reference's own mlefcn at a few interior points (gradient pin)
    pts = np.array([[0.1036165, 0.0995048], [0.3, 0.2], [0.8, 1.1]])
    h = 1e-6
    fd = np.zeros((len(pts), 2, 2))
    for si, s in enumerate([0.0, 0.05]):
        for pi, pt in enumerate(pts):
            for k in range(2):
                e = np.zeros(2)
                e[k] = h
                fp = np.squeeze(v4['mlefcn'](xT, yT, Daa, pt + e, s))
                fm = np.squeeze(v4['mlefcn'](xT, yT, Daa, pt - e, s))
                fd[pi, k, si] = (fp - fm) / (2 * h)
    save(a.out, 'g6_mle_grid', dict(xTrain=xT, yTrain=yT, th0=th0, nll_sigma0=M, nll_sigma005=M2,
                                    fd_points=pts, fd_sigmas=np.array([0.0, 0.05]), fd_grad=fd, fd_h=np.array(h)))

    # ---- G2: 1-D synthetic BO function (BayesOpt_v4.ipynb:248-250)
    P = 400
    x1 = np.linspace(-5, 5, P)[:, np.newaxis]
    y1 = np.exp(-x1 ** 2) + np.sin(2 * x1) * np.cos(10 * x1)
    rng = np.random.default_rng(2)
    for nT in (3, 8, 20):
        ii = rng.permutation(P)[:nT]
        for s in (0.0, 1e-2):
            th = np.array([0.25 + 0.05 * nT / 20])
            g = run_gprfcn(v4, x1, x1[ii], y1[ii], th, s, 0.01, keep_factor=True)
            save(a.out, 'g2_1d_nT%d_s%s' % (nT, '0' if s == 0 else '1e-2'), g)

    # ---- G3: synthetic d=4
    for n in (64, 512):
        for s in (0.0, 1e-2, 1e-1):
            rng = np.random.default_rng(300 + n + int(s * 1000))
            X = rng.random((n, 4))
            yv = (np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n))[:, None]
            Xs = rng.random((2000, 4))
            th = rng.uniform(0.3, 0.8, 4)
            g = run_gprfcn(v4, Xs, X, yv, th, s, 0.01, keep_factor=(n == 64))
            save(a.out, 'g3_d4_n%d_s%s' % (n, {0.0: '0', 1e-2: '1e-2', 1e-1: '1e-1'}[s]), g)

    # ---- G4: d=8, n=1024, m=1500, sigmaY=1e-3
    rng = np.random.default_rng(4)
    X = rng.random((1024, 8))
    yv = (np.cos(2 * X[:, :3].sum(1)) + X[:, 5] ** 2)[:, None]
    Xs = rng.random((1500, 8))
    g = run_gprfcn(v4, Xs, X, yv, np.full(8, 0.9), 1e-3, 0.01, keep_factor=False)
    save(a.out, 'g4_d8_n1024', g)

    # ---- G5: degenerate inputs
    rng = np.random.default_rng(5)
    X = rng.random((40, 3))
    yv = np.sin(4 * X[:, 0:1]) + X[:, 1:2]
    Xs = np.vstack([X[:10], rng.random((200, 3))])       # first 10 candidates ARE training points
    g = run_gprfcn(v4, Xs, X, yv, np.array([0.5, 0.6, 0.7]), 0.0, 0.01, keep_factor=True)
    save(a.out, 'g5_cand_equals_train', g)
    Xd = np.vstack([X, X[:3]])                           # duplicate training rows, sigmaY=0
    yd = np.vstack([yv, yv[:3]])
    g = run_gprfcn(v4, Xs, Xd, yd, np.array([0.5, 0.6, 0.7]), 0.0, 0.01, keep_factor=False)
    save(a.out, 'g5_dup_train_s0', g)
    g = run_gprfcn(v4, Xs, Xd, yd, np.array([0.5, 0.6, 0.7]), 1e-2, 0.01, keep_factor=False)
    save(a.out, 'g5_dup_train_s1e-2', g)

    # ---- G7: kriging (Gaussian_Process_Regression.ipynb:197-203, 228-234; usage :514-525, :678-693)
    s = np.linspace(-5, 5, 1200)
    Y = np.exp(-s ** 2) + np.cos(2 * s) * np.sin(2 * s)
    rng = np.random.default_rng(7)
    ii = rng.permutation(1200)[:12]
    xT1, yT1 = s[ii][:, None], Y[ii]
    s2 = np.linspace(-5, 5, 150)
    for deg in (0, 1):
        theta = np.array([0.8])
        beta, cov, K, Ks, Kss, FT, FS = gpr['gpr_poly'](xT1, yT1, xT1, theta, 'gaussian', deg)
        mu = np.zeros(len(s2))
        mse = np.zeros(len(s2))
        for i in range(len(s2)):
            aa = np.array([s2[i]])[np.newaxis, :]
            u, v = gpr['y_kriging_v2'](aa, beta, cov, K, FT, xT1, yT1, theta, deg, 'gaussian')
            mu[i], mse[i] = np.squeeze(u), np.squeeze(v)
        pi = gpr['acquisition'](mu, yT1, np.sqrt(np.abs(mse)) + 1e-300)
        save(a.out, 'g7_kriging_1d_deg%d' % deg, dict(xT=xT1, yT=yT1, xS=s2[:, None], theta=theta, deg=np.array(deg),
                                                      beta=np.asarray(beta), cov=np.array(float(np.squeeze(cov))),
                                                      mu=mu, mse=mse, pi=pi))
    g2 = np.linspace(-2, 2, 20)
    XX = np.array([[u, v] for u in g2 for v in g2])
    YY = np.sin(XX[:, 0]) * np.cos(XX[:, 1]) + 0.3 * XX[:, 0]
    ii = rng.permutation(400)[:20]
    xT2, yT2 = XX[ii], YY[ii]
    xS2 = rng.uniform(-2, 2, (120, 2))
    for deg in (0, 1):
        theta = np.array([0.9])
        beta, cov, K, Ks, Kss, FT, FS = gpr['gpr_poly'](xT2, yT2, xT2, theta, 'gaussian', deg)
        mu = np.zeros(len(xS2))
        mse = np.zeros(len(xS2))
        for i in range(len(xS2)):
            u, v = gpr['y_kriging_v2'](xS2[i][None, :], beta, cov, K, FT, xT2, yT2, theta, deg, 'gaussian')
            mu[i], mse[i] = np.squeeze(u), np.squeeze(v)
        save(a.out, 'g7_kriging_2d_deg%d' % deg, dict(xT=xT2, yT=yT2, xS=xS2, theta=theta, deg=np.array(deg),
                                                      beta=np.asarray(beta), cov=np.array(float(np.squeeze(cov))),
                                                      mu=mu, mse=mse))


if __name__ == '__main__':
    main()
