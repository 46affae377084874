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


def load_cell_defining(path, name, ns):
    """exec the code cell of notebook `path` that defines function `name` into namespace ns."""
    nb = json.load(open(path))
    for c in nb['cells']:
        src = ''.join(c['source'])
        if c['cell_type'] == 'code' and ('def ' + name + '(') in src:
            exec(compile(src, path, 'exec'), ns)
            return ns
    raise RuntimeError(name + ' not found in ' + path)


def lattice_loop(a, v4):
    """G8: the body of the 1-D BO loop (BayesOpt_v4.ipynb:642-664) on the notebook's own equispaced grid
    (x = np.linspace(-5, 5, 400), :248), theta held fixed (the optimiser needs autograd), ten iterations: the sequence of
    picked indices and max EI, plus the gap between the two largest EI values of every iteration (how close a tie was)."""
    P = 400
    x = np.linspace(-5, 5, P)[:, np.newaxis]
    y = np.exp(-x ** 2) + np.sin(2 * x) * np.cos(10 * x)
    for tag, s in (('s0', 0.0), ('s1e-2', 1e-2)):
        ii = [37, 180, 333]
        xT, yT = x[ii], y[ii]
        theta = np.array([0.3])
        jj_hist, ei_hist, gap_hist, mumax_hist, EI, MU, VAR = [], [], [], [], [], [], []
        for it in range(10):
            Mu, Cov, _, _, _, _, _, _ = v4['gprfcn'](x, xT, yT, theta, s)
            ei = v4['expected_improvement'](Mu, Cov, 0.01)
            jj = int(np.argmax(ei))
            top2 = np.sort(ei)[-2:]
            jj_hist.append(jj)
            ei_hist.append(float(ei[jj]))
            gap_hist.append(float(top2[1] - top2[0]))
            mumax_hist.append(float(np.max(Mu)))
            EI.append(ei.copy())
            MU.append(np.squeeze(Mu).copy())
            VAR.append(np.diag(Cov).copy())
            xT = np.vstack((xT, x[jj, :]))
            yT = np.vstack((yT, y[jj]))
        save(a.out, 'g8_lattice_loop_' + tag, dict(x=x, y=y, start=np.array(ii), theta=theta, sigmaY=np.array(s), kappa=np.array(0.01),
                                                   jj=np.array(jj_hist), ei_max=np.array(ei_hist), top2_gap=np.array(gap_hist),
                                                   mu_max=np.array(mumax_hist), EI=np.array(EI), Mu=np.array(MU), var=np.array(VAR)))


def descent_paths(a, v4, g1_train):
    """G9: the reference's own conj_gradient_descent (BayesOpt_v4.ipynb:31-87) executed with `grad` (autograd, absent
    here) replaced by the analytic gradient of the CPU restatement (oracle/gp_oracle.py:mle_grad, itself pinned by
    central differences of the reference's mlefcn in g6).  g is the reference's mlefcn, exactly as hypopt builds it (:168)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle import gp_oracle as o
    xT, yT = g1_train
    sY = 0.05
    Daa = v4['distmat_v2'](xT, xT)
    mle = lambda w: float(np.squeeze(v4['mlefcn'](xT, yT, Daa, w, sY)))
    ns = {'np': np, 'grad': lambda g: (lambda w: o.mle_grad(xT, yT, w, sY))}
    load_cell_defining(os.path.join(a.ref, 'BayesOpt', 'BayesOpt_v4.ipynb'), 'conj_gradient_descent', ns)
    cgd = ns['conj_gradient_descent']
    x0 = 0.08 + 0.5 * o.splitmix64_uniform(33, 4 * 2).reshape(4, 2)
    out = dict(xTrain=xT, yTrain=yT, sigmaY=np.array(sY), x0=x0)
    cases = [('component_normal', 0.01, 1, 'component', 'normal', 40), ('component_momentum', 0.01, 0.6, 'component', 'normal', 40),
             ('full_backtrack', 0.4, 1, 'full', 'backtrack', 6), ('full_diminishing', 0.02, 1, 'full', 'diminishing', 12),
             ('plain_normal', 1e-5, 1, 'none', 'normal', 25)]
    for tag, alpha, beta, norm, step, its in cases:
        W = np.zeros((len(x0), its + 1, 2))
        Cst = np.zeros((len(x0), its + 1))
        flags = np.zeros(len(x0), dtype=int)
        for r in range(len(x0)):
            wh, ch, fl = cgd(mle, alpha, beta, norm, step, its, x0[r].copy())
            assert len(wh) == its + 1, (tag, len(wh))
            W[r] = np.array(wh)
            Cst[r] = np.array(ch)
            flags[r] = fl
        out[tag + '_w'] = W
        out[tag + '_cost'] = Cst
        out[tag + '_flag'] = flags
        out[tag + '_args'] = np.array([alpha, beta, its])
    save(a.out, 'g9_descent_paths', out)


def kriging_more(a, gpr):
    """More of G7: trend degrees 2 and 3 and three input columns (polymat, Gaussian_Process_Regression.ipynb:113-140), and a
    nearly singular K on which the reference's LU solves (np.linalg.solve, :201-202) still return numbers."""
    rng = np.random.default_rng(77)

    def run(name, xT, yT, xS, theta, deg):
        beta, cov, K, Ks, Kss, FT, FS = gpr['gpr_poly'](xT, yT, xT, theta, 'gaussian', deg)
        mu = np.zeros(len(xS))
        mse = np.zeros(len(xS))
        for i in range(len(xS)):
            u, v = gpr['y_kriging_v2'](xS[i][None, :], beta, cov, K, FT, xT, yT, theta, deg, 'gaussian')
            mu[i], mse[i] = np.squeeze(u), np.squeeze(v)
        save(a.out, name, dict(xT=xT, yT=yT, xS=xS, theta=theta, deg=np.array(deg), beta=np.asarray(beta),
                               cov=np.array(float(np.squeeze(cov))), mu=mu, mse=mse, condK=np.array(np.linalg.cond(K))))

    s = np.linspace(-5, 5, 1200)
    Y = np.exp(-s ** 2) + np.cos(2 * s) * np.sin(2 * s)
    ii = rng.permutation(1200)[:14]
    run('g7_kriging_1d_deg2', s[ii][:, None], Y[ii], np.linspace(-5, 5, 90)[:, None], np.array([0.8]), 2)
    g2 = np.linspace(-2, 2, 20)
    XX = np.array([[u, v] for u in g2 for v in g2])
    YY = np.sin(XX[:, 0]) * np.cos(XX[:, 1]) + 0.3 * XX[:, 0] + 0.1 * XX[:, 0] * XX[:, 1]
    ii = rng.permutation(400)[:24]
    xS2 = rng.uniform(-2, 2, (100, 2))
    for deg in (2, 3):
        run('g7_kriging_2d_deg%d' % deg, XX[ii], YY[ii], xS2, np.array([0.9]), deg)
    X3 = rng.uniform(-1, 1, (40, 3))
    Y3 = np.sin(2 * X3[:, 0]) + X3[:, 1] * X3[:, 2] + 0.5 * X3[:, 2] ** 2
    xS3 = rng.uniform(-1, 1, (80, 3))
    for deg in (0, 1, 2, 3):
        run('g7_kriging_3d_deg%d' % deg, X3, Y3, xS3, np.array([0.7]), deg)
    # nearly singular: 60 close points, long length scale -> cond(K) ~ 1e17+; LU with pivoting returns finite numbers
    xs = np.linspace(0, 1, 60)[:, None]
    ys = np.sin(3 * xs[:, 0])
    run('x7_kriging_singular', xs, ys, np.linspace(0, 1, 25)[:, None] + 0.003, np.array([2.0]), 0)


def gpr_notebook_cells(a, gpr):
    """GPRNB: the inputs of the GPR notebook's own driver cells, with a seeded permutation in place of the unseeded one:
    1-D, 1200-point curve, nT = 3, deg 0, epsilon 1e-9 (Gaussian_Process_Regression.ipynb:372-388, 405-445, 473-526);
    2-D, 20 x 20 grid on [-pi, pi]^2, every nsamp-th point of a permutation, deg 1, epsilon 1e-8 (:551-592, 621-694);
    3-D, 800 of the 20^3 grid points on [-1, 1]^3, deg 0, theta = 1 (:811-874).  Stored: what gpr_poly / y_kriging_v2 /
    y_kriging + mse_kriging / gpr_mle return, cond(K), and the path of the notebook's descent on gpr_mle with central
    differences of gpr_mle itself standing in for autograd's gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    ns = {'np': np, 'grad': None}
    load_cell_defining(os.path.join(a.ref, 'BayesOpt', 'Gaussian_Process_Regression.ipynb'), 'conj_gradient_descent', ns)

    def fd_grad(fun, h=1e-6):
        return lambda w: np.array([(float(np.squeeze(fun(w + h * np.abs(w)))) - float(np.squeeze(fun(w - h * np.abs(w))))) / (2 * h * abs(float(w[0])))])

    def predict(xT, yT, xS, theta, deg):
        beta, cov, K, Ks, Kss, FT, FS = gpr['gpr_poly'](xT, yT, xT, theta, 'gaussian', deg)
        mu, mse = np.zeros(len(xS)), np.zeros(len(xS))
        for i in range(len(xS)):
            u, v = gpr['y_kriging_v2'](xS[i][None, :], beta, cov, K, FT, xT, yT, theta, deg, 'gaussian')
            mu[i], mse[i] = np.squeeze(u), np.squeeze(v)
        return np.asarray(beta), float(np.squeeze(cov)), K, FT, mu, mse

    def mle_grid(xT, yT, thetas, deg, eps):
        out = np.zeros(len(thetas))
        for i, t in enumerate(thetas):
            out[i] = float(np.squeeze(gpr['gpr_mle'](np.array([t]), xT, yT, xT, 'gaussian', deg, eps)))
        return out

    def descent(xT, yT, deg, eps, w0s, max_its):
        fun = lambda w: gpr['gpr_mle'](w, xT, yT, xT, 'gaussian', deg, eps)
        W, Cst = np.zeros((len(w0s), max_its)), np.zeros((len(w0s), max_its))
        for r, w0 in enumerate(w0s):
            ns['grad'] = lambda g: fd_grad(g)
            wh, ch = ns['conj_gradient_descent'](fun, 0.01, 0.5, 'component', 'standard', max_its, np.array([w0]))
            W[r] = np.squeeze(np.array(wh))
            Cst[r] = np.squeeze(np.array(ch))
        return W, Cst

    # ---- 1-D (:372-388): nT = 3 of 1200 points
    s = np.linspace(-5, 5, 1200)
    Y = np.exp(-s ** 2) + np.cos(2 * s) * np.sin(2 * s)
    inds = np.random.default_rng(1001).permutation(1200)
    xT, yT = s[:, None][inds[:3]], Y[inds[:3]]
    xS = np.linspace(-5, 5, 1500)[::10][:, None]
    for theta in (0.5, 1.0):
        beta, cov, K, FT, mu, mse = predict(xT, yT, xS, np.array([theta]), 0)
        tg = np.geomspace(0.15, 3.0, 12)
        w0s = np.array([0.35, 0.62, 0.9])
        W, Cst = descent(xT, yT, 0, 1e-9, w0s, 100)
        save(a.out, 'gprnb_1d_theta%s' % ('05' if theta == 0.5 else '1'),
             dict(xT=xT, yT=yT, xS=xS, theta=np.array([theta]), deg=np.array(0), beta=beta, cov=np.array(cov), mu=mu, mse=mse,
                  condK=np.array(np.linalg.cond(K)), mle_thetas=tg, mle_eps=np.array(1e-9), mle=mle_grid(xT, yT, tg, 0, 1e-9),
                  descent_w0=w0s, descent_w=W, descent_cost=Cst, descent_args=np.array([0.01, 0.5, 100])))

    # ---- 2-D (:551-592): 20 x 20 grid on [-pi, pi]^2, theta = 1.05, deg 1
    N = 20
    s = np.linspace(-np.pi, np.pi, N)
    xp = np.array([[s[i], s[j]] for i in range(N) for j in range(N)])
    yp = (2.5 * np.exp(-(xp[:, 0] ** 2 + xp[:, 1] ** 2)) + np.sin(2 * xp[:, 0]) + np.cos(xp[:, 1]))[:, None]
    inds = np.random.default_rng(1002).permutation(N * N)
    s2 = np.linspace(-np.pi, np.pi, 50)
    xS = np.array([[s2[i], s2[j]] for i in range(0, 50, 3) for j in range(0, 50, 3)])
    for nsamp in (50, 20, 10, 5, 2):
        I = inds[::nsamp]
        xT, yT = xp[I], yp[I]
        theta = np.array([1.05])
        beta, cov, K, FT, mu, mse = predict(xT, yT, xS, theta, 1)
        d = dict(xT=xT, yT=yT, xS=xS, theta=theta, deg=np.array(1), beta=beta, cov=np.array(cov), mu=mu, mse=mse,
                 condK=np.array(np.linalg.cond(K)))
        if nsamp >= 10:
            tg = np.geomspace(0.2, 2.0, 10)
            d.update(mle_thetas=tg, mle_eps=np.array(1e-8), mle=mle_grid(xT, yT, tg, 1, 1e-8))
        if nsamp == 20:
            w0s = np.array([0.5, 1.3])
            W, Cst = descent(xT, yT, 1, 1e-8, w0s, 120)
            d.update(descent_w0=w0s, descent_w=W, descent_cost=Cst, descent_args=np.array([0.01, 0.5, 120]))
        save(a.out, 'gprnb_2d_nsamp%d' % nsamp, d)

    # ---- 3-D (:811-874): 800 of the 20^3 grid points on [-1, 1]^3, theta = 1, deg 0; y_kriging / mse_kriging as the cell calls them
    s = np.linspace(-1, 1, N)
    xp = np.array([[s[i], s[j], s[k]] for i in range(N) for j in range(N) for k in range(N)])
    yp = (np.exp(-(xp ** 2).sum(1) / 10) + np.sin(2 * xp[:, 0]) + np.cos(xp[:, 1]) + 6 * np.cos(xp[:, 2]))[:, None]
    inds = np.random.default_rng(1003).permutation(N ** 3)
    I = inds[::10]
    xT, yT = xp[I], yp[I]
    s2 = np.linspace(-1, 1, 30)
    xS = np.array([[s2[i], s2[j], s2[k]] for i in range(0, 30, 6) for j in range(0, 30, 6) for k in range(0, 30, 6)])
    theta = np.array([1.0])
    beta, cov, K, FT, mu, mse = predict(xT, yT, xS, theta, 0)
    gpr['cov'] = cov                                     # y_kriging reads a global named cov (:224)
    mu_w, mse_w = np.zeros(len(xS)), np.zeros(len(xS))
    for i in range(len(xS)):
        u, _ = gpr['y_kriging'](xS[i][None, :], beta, K, FT, xT, yT, theta, 0, 'gaussian')
        mu_w[i] = np.squeeze(u)
        mse_w[i] = np.squeeze(gpr['mse_kriging'](xS[i][None, :], cov, K, FT, xT, yT, theta, 0, 'gaussian'))
    save(a.out, 'gprnb_3d_n800', dict(xT=xT, yT=yT, xS=xS, theta=theta, deg=np.array(0), beta=beta, cov=np.array(cov), mu=mu,
                                          mse=mse, mu_w=mu_w, mse_w=mse_w, condK=np.array(np.linalg.cond(K))))
    # the same two functions where K is well conditioned (40 of the points, w0 = 0.09 <-> theta = 0.3)
    I = inds[:40]
    xT, yT = xp[I], yp[I]
    theta = np.array([0.3])
    beta, cov, K, FT, mu, mse = predict(xT, yT, xS, theta, 0)
    gpr['cov'] = cov
    w0 = np.array([0.2])
    mu_w, mse_w = np.zeros(len(xS)), np.zeros(len(xS))
    for i in range(len(xS)):
        u, _ = gpr['y_kriging'](xS[i][None, :], beta, K, FT, xT, yT, w0, 0, 'gaussian')
        mu_w[i] = np.squeeze(u)
        mse_w[i] = np.squeeze(gpr['mse_kriging'](xS[i][None, :], cov, K, FT, xT, yT, w0, 0, 'gaussian'))
    save(a.out, 'gprnb_3d_n40', dict(xT=xT, yT=yT, xS=xS, theta=theta, w0=w0, deg=np.array(0), beta=beta, cov=np.array(cov), mu=mu,
                                         mse=mse, mu_w=mu_w, mse_w=mse_w, condK=np.array(np.linalg.cond(K))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default='', help='comma-separated groups to (re)write, e.g. g8,g9,g7more; default: everything')
    a = ap.parse_args()
    only = set(filter(None, a.only.split(',')))
    os.makedirs(a.out, exist_ok=True)
    warnings.simplefilter('ignore')
    v4 = load_function_cell(os.path.join(a.ref, 'BayesOpt', 'BayesOpt_v4.ipynb'),
                            ['gprfcn', 'mlefcn', 'expected_improvement'])
    gpr = load_function_cell(os.path.join(a.ref, 'BayesOpt', 'Gaussian_Process_Regression.ipynb'),
                             ['gpr_poly', 'y_kriging_v2', 'polymat', 'gpr_mle', 'y_kriging', 'mse_kriging'])

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
    if not only or 'g1' in only:
        save(a.out, 'g1_mat_mixture', g1)
    if not only or 'g8' in only:
        lattice_loop(a, v4)
    if not only or 'g9' in only:
        descent_paths(a, v4, (xT, yT))
    if not only or 'g7more' in only:
        kriging_more(a, gpr)
    if not only or 'gprnb' in only:
        gpr_notebook_cells(a, gpr)
    if only and not (only & {'g2', 'g3', 'g4', 'g5', 'g6', 'g7'}):
        return

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
