#!/usr/bin/env python3
"""CPU oracle at the size of the 64 M-candidate configuration (n = 32768, d = 8, Matern-5/2): TEST INFRASTRUCTURE.

The reference cannot run this size (it has no Matern-5/2 kernel and forms m x m matrices), so the outputs below come from
the CPU restatement (oracle/gp_oracle.py): its kernel functions, its predict(); only the fit is re-stated memory-lean here
(Fortran-ordered storage factored in place by LAPACK dpotrf, two n x n matrices instead of six) and checked against
oracle.fit at a small size before it is trusted.  Takes ~10 minutes and ~20 GB on 8 cores.

    python oracle/make_oracle_fixtures.py fit                      # fit once; factor cached under /tmp; writes tests/golden/o1_cfg3_first512.npz
    python oracle/make_oracle_fixtures.py compare <sample.npz>     # oracle on the sample of scripts/full_cfg3.py -> profiles/r2_cfg3_full_parity.json
    python oracle/make_oracle_fixtures.py grad                     # -logL + analytic gradient at n = 16384 for two settings of the
                                                                   # 256-setting sweep -> tests/golden/o2_cfg5_grad.npz (~10 min, ~20 GB)
"""
import json
import math
import os
import sys
import time

import numpy as np
from scipy.linalg import solve_triangular

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as o  # noqa: E402

CFG3 = dict(n=32768, d=8, kernel=o.MATERN52, theta=0.6, sigmaY=0.1, kappa=0.01, seed_x=3, seed_c=33)
CACHE = '/tmp/cfg3_oracle'


def training_set(cfg):
    n, d = cfg['n'], cfg['d']
    X = o.splitmix64_uniform(cfg['seed_x'], n * d).reshape(n, d)
    u1 = o.splitmix64_uniform(cfg['seed_x'] + 100, n)
    u2 = o.splitmix64_uniform(cfg['seed_x'] + 100 + 7919, n)
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * (np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2))
    return X, y, np.full(d, cfg['theta'])


def candidates(cfg, rows):
    """Rows (any index array) of the configuration's candidate stream."""
    d = cfg['d']
    rows = np.asarray(rows, dtype=np.int64)
    out = np.empty((len(rows), d))
    for k, r in enumerate(rows):
        out[k] = o.splitmix64_uniform(cfg['seed_c'], d, offset=int(r) * d)
    return out


def chol_lower(A, pb):
    """Lower Cholesky factor of A, overwriting its lower triangle (the strict upper triangle keeps stale values that no
    triangular solve reads): right-looking over panels of pb columns, np.linalg.cholesky on the diagonal block, a
    triangular solve for the panel, GEMMs for the trailing update.  Stands in for one np.linalg.cholesky call because
    OpenBLAS 0.3.29 (numpy's and scipy's builds alike) segfaults inside dpotrf at n = 32768 in this container."""
    n = A.shape[0]
    for p in range(0, n, pb):
        e = min(n, p + pb)
        Lpp = np.linalg.cholesky(A[p:e, p:e])
        A[p:e, p:e] = Lpp
        if e < n:
            A[e:, p:e] = solve_triangular(Lpp, A[e:, p:e].T, lower=True).T
            for q in range(e, n, 4096):          # trailing update in row slabs (bounded temporaries), lower part only
                f = min(n, q + 4096)
                A[q:f, e:f] -= A[q:f, p:e] @ A[e:f, p:e].T
    return A


def lean_fit_v4(X, y, theta, sigmaY, kernel, blk=2048, pb=4096):
    """o.fit(variant=V4) with fewer n x n temporaries: the same kernel functions (block by block), the same jitter rule, the
    same np.linalg.cholesky / solve_triangular calls.  (scipy's own LAPACK segfaults in dpotrf at n = 32768; numpy's does not.)"""
    n = X.shape[0]
    eps = o.jitter(n, sigmaY)
    A = np.empty((n, n))
    for lo in range(0, n, blk):
        hi = min(n, lo + blk)
        A[lo:hi, :] = o.kernel_matrix(X[lo:hi], X, theta, kernel)
    A[np.diag_indices(n)] += eps
    K0 = None
    if n > 16384:
        K0 = A.copy()
        L1 = chol_lower(A, pb)                                    # K0 + eps I, in place
    else:
        L1 = np.linalg.cholesky(A)
    a1 = solve_triangular(L1, y, lower=True)
    a1 = solve_triangular(L1, a1, lower=True, trans='T')
    theta1 = float(y @ a1) / n
    nll = n / 2 * (math.log(2 * math.pi) + 1) + n / 2 * math.log(theta1) + float(np.sum(np.log(np.diag(L1))))
    del L1
    if K0 is not None:
        A = K0
    A[np.diag_indices(n)] = 1.0                                   # back to K0: k(x, x) = 1 exactly for both kernels
    A *= theta1
    A[np.diag_indices(n)] += eps
    L2 = chol_lower(A, pb) if n > 16384 else np.linalg.cholesky(A)   # theta1 K0 + eps I
    return dict(X=X, y=y, theta=theta, sigmaY=float(sigmaY), kernel=kernel, variant=o.VARIANT_V4, n=n, L2=L2, alpha1=a1,
                alpha=a1 / theta1, theta1=theta1, nll=nll, eps=eps)


def self_check():
    cfg = dict(CFG3, n=700)
    X, y, th = training_set(cfg)
    a = lean_fit_v4(X, y, th, cfg['sigmaY'], cfg['kernel'], blk=256)
    b = o.fit(X, y, th, cfg['sigmaY'], kernel=cfg['kernel'])
    Xs = candidates(cfg, np.arange(300))
    ma, va = o.predict(a, Xs)
    mb, vb = o.predict(b, Xs)
    assert abs(a['theta1'] - b['theta1']) <= 1e-13 * b['theta1'] and abs(a['nll'] - b['nll']) <= 1e-12 * abs(b['nll'])
    assert np.max(np.abs(ma - mb)) <= 1e-12 * np.max(np.abs(mb)) and np.max(np.abs(va - vb)) <= 1e-12 * np.max(np.abs(vb))
    K = o.kernel_matrix(X, X, th, cfg['kernel']) + 0.01 * np.eye(700)
    Lb = np.tril(chol_lower(K.copy(), 256))
    assert np.max(np.abs(Lb - np.linalg.cholesky(K))) <= 1e-13
    print('lean fit == oracle.fit at n=700; blocked factor == np.linalg.cholesky')


def do_fit():
    self_check()
    X, y, th = training_set(CFG3)
    t0 = time.time()
    st = lean_fit_v4(X, y, th, CFG3['sigmaY'], CFG3['kernel'])
    print('fit %.0f s, theta1 %.15g, nll %.15g' % (time.time() - t0, st['theta1'], st['nll']), flush=True)
    np.save(CACHE + '_L2.npy', st['L2'])
    np.savez(CACHE + '_vec.npz', alpha=st['alpha'], alpha1=st['alpha1'], theta1=st['theta1'], nll=st['nll'])
    rows = np.arange(512)
    mu, var = o.predict(st, candidates(CFG3, rows), block=512)
    ei = o.expected_improvement(mu, var, CFG3['kappa'])
    out = os.path.join(ROOT, 'tests', 'golden', 'o1_cfg3_first512.npz')
    np.savez_compressed(out, rows=rows, mu=mu, var=var, ei_local_incumbent=ei, theta1=np.array(st['theta1']), nll=np.array(st['nll']),
                        alpha_head=st['alpha'][:64], config=np.array(json.dumps(CFG3)), numpy_version=np.array(np.__version__),
                        source=np.array('oracle/make_oracle_fixtures.py (CPU restatement; no reference implementation of Matern-5/2)'))
    print('wrote', out)


def load_state():
    X, y, th = training_set(CFG3)
    v = np.load(CACHE + '_vec.npz')
    L2 = np.load(CACHE + '_L2.npy', mmap_mode=None)
    return dict(X=X, y=y, theta=th, sigmaY=CFG3['sigmaY'], kernel=CFG3['kernel'], variant=o.VARIANT_V4, n=CFG3['n'], L2=L2,
                alpha=v['alpha'], alpha1=v['alpha1'], theta1=float(v['theta1']), nll=float(v['nll']))


def do_compare(sample_path):
    s = np.load(sample_path)
    st = load_state()
    rows = np.concatenate([s['sub'], s['top_index']])
    mu, var = o.predict(st, candidates(CFG3, rows), block=1024)
    mu_max = float(s['mu_max'])                      # the incumbent is the maximum over all 64 M candidates: taken from the GPU run
    ei = o.expected_improvement(mu, var, CFG3['kappa'], incumbent=mu_max)
    ns = len(s['sub'])
    rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
    top_ei_oracle = ei[ns:]
    order = np.lexsort((s['top_index'], -top_ei_oracle))
    res = {
        'what': 'GPU full-size run (scripts/full_cfg3.py) against the CPU oracle at n=32768 on the strided subsample and the 16 largest-EI candidates',
        'candidates_total': int(s['m']), 'subsample': ns,
        'mean_relmax': rel(s['mu'], mu[:ns]), 'var_relmax': rel(s['var'], var[:ns]), 'ei_relmax': rel(s['ei'], ei[:ns]),
        'theta1_rel': abs(float(s['theta1']) - st['theta1']) / st['theta1'], 'nll_rel': abs(float(s['nll']) - st['nll']) / abs(st['nll']),
        'top16_mean_relmax': rel(s['top_mu'], mu[ns:]), 'top16_var_relmax': rel(s['top_var'], var[ns:]), 'top16_ei_relmax': rel(s['top_ei'], top_ei_oracle),
        'gpu_argmax': int(s['top_index'][0]), 'oracle_argmax_among_gpu_top16': int(s['top_index'][order[0]]),
        'argmax_agrees': bool(int(s['top_index'][order[0]]) == int(s['top_index'][0])),
        'gpu_top2_ei_gap': float(s['top_ei'][0] - s['top_ei'][1]), 'oracle_top2_ei_gap': float(top_ei_oracle[order[0]] - top_ei_oracle[order[1]]),
        'tolerances': {'mean': 1e-10, 'var': 1e-8},
    }
    out = os.path.join(ROOT, 'profiles', 'r2_cfg3_full_parity.json')
    with open(out, 'w') as f:
        f.write(json.dumps(res, indent=1) + '\n')
    print(json.dumps(res, indent=1))


def do_grad():
    """The likelihood sweep's data (n = 16384, d = 8, RBF) and two of its 256 (length-scale multiplier, sigmaY) settings: -logL
    (o.mlefcn, BayesOpt_v4.ipynb:154-165) and its analytic gradient (o.mle_grad, GP/BayesOpt_v3.m:392-417) from the CPU
    restatement, called as they are.  A third, small problem is evaluated the same way and compared with central differences
    of o.mlefcn first."""
    Xs_, ys_ = o.splitmix64_uniform(9, 300 * 3).reshape(300, 3), None
    ys_ = np.sin(3 * Xs_.sum(1))
    th_ = np.array([0.4, 0.5, 0.6])
    g_ = o.mle_grad(Xs_, ys_, th_, 0.1)
    for t in range(3):
        e = np.zeros(3)
        e[t] = 1e-6
        fd = (o.mlefcn(Xs_, ys_, th_ + e, 0.1) - o.mlefcn(Xs_, ys_, th_ - e, 0.1)) / 2e-6
        assert abs(fd - g_[t]) <= 1e-6 * np.max(np.abs(g_)), (t, fd, g_[t])
    n, d = 16384, 8
    X = o.splitmix64_uniform(5, n * d).reshape(n, d)
    u1, u2 = o.splitmix64_uniform(105, n), o.splitmix64_uniform(105 + 7919, n)
    y = np.sin(3 * X.sum(1)) + 0.1 * (np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2))
    mult = np.geomspace(0.25, 2.0, 16)
    sigs = np.geomspace(1e-2, 1.0, 16)
    picks = [(5, 8), (11, 4)]                                   # (multiplier index, sigma index) in the 16 x 16 grid
    thetas = np.array([np.full(d, 0.5 * mult[i]) * (1 + 0.05 * np.arange(d)) for i, _ in picks])   # distinct scales per dimension
    sg = np.array([sigs[j] for _, j in picks])
    nll, grad = np.zeros(len(picks)), np.zeros((len(picks), d))
    for b in range(len(picks)):
        t0 = time.time()
        nll[b] = o.mlefcn(X, y, thetas[b], float(sg[b]))
        grad[b] = o.mle_grad(X, y, thetas[b], float(sg[b]))
        print('setting %d: nll %.15g  |grad| %.6g  (%.0f s)' % (b, nll[b], np.max(np.abs(grad[b])), time.time() - t0), flush=True)
    out = os.path.join(ROOT, 'tests', 'golden', 'o2_cfg5_grad.npz')
    np.savez_compressed(out, n=np.array(n), d=np.array(d), seed_x=np.array(5), seed_noise=np.array(105), thetas=thetas, sigmas=sg,
                        nll=nll, grad=grad, picks=np.array(picks), numpy_version=np.array(np.__version__),
                        source=np.array('oracle/make_oracle_fixtures.py grad (CPU restatement: mlefcn + mle_grad as they are)'))
    print('wrote', out)


if __name__ == '__main__':
    if len(sys.argv) >= 2 and sys.argv[1] == 'fit':
        do_fit()
    elif len(sys.argv) >= 2 and sys.argv[1] == 'grad':
        do_grad()
    elif len(sys.argv) >= 3 and sys.argv[1] == 'compare':
        do_compare(sys.argv[2])
    else:
        print(__doc__)
