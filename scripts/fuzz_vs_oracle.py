#!/usr/bin/env python3
"""Randomised differential run: GPU library against the CPU oracle over random shapes, kernels, variants and noise levels
(well-conditioned settings only, where the 1e-10 / 1e-8 targets apply).  python3 scripts/fuzz_vs_oracle.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP
from oracle import gp_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
gp = GP(0)
worst = dict(mu=0, var=0, ei=0, nll=0, th1=0)
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 17, 64, 127, 128, 129, 255, 256, 257, 300, 511, 513, 640, 900]))
    d = int(rng.integers(1, 11))
    m = int(rng.choice([1, 2, 63, 255, 256, 257, 1000, 4097, 9001]))
    kernel = int(rng.integers(0, 2))
    variant = int(rng.choice([bo.VARIANT_V4, bo.VARIANT_SINGLE, bo.VARIANT_M3]))
    sY = float(rng.choice([0.05, 0.1, 0.3]))
    theta = rng.uniform(0.3, 0.9, d) * (1 + 0.15 * d)
    X = rng.random((n, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n); Xs = rng.random((m, d))
    chunk = int(rng.choice([0, 256, 1024]))
    gp.set_chunk(chunk)
    gp.fit(X, y, theta, sY, kernel=kernel, variant=variant)
    mu, var, ei, idx, ei_max, mu_max = gp.predict_ei(Xs, 0.01)
    mu_o, var_o, st = o.gprfcn_diag(Xs, X, y, theta, sY, kernel=kernel, variant=variant)
    ei_o = o.expected_improvement(mu_o, var_o, 0.01)
    e = dict(mu=rel(mu, mu_o), var=rel(var, var_o), ei=float(np.max(np.abs(ei - ei_o)) / max(np.max(np.abs(ei_o)), 1e-300)),
             nll=abs(gp.nll - st['nll']) / abs(st['nll']), th1=abs(gp.theta1 - st['theta1']) / st['theta1'])
    for k in worst:
        worst[k] = max(worst[k], e[k])
    top2 = np.sort(ei_o)[-2:] if m > 1 else np.array([0, 1.0])
    tie = (top2[1] - top2[0]) <= 1e-9 * abs(top2[1])
    cond = float(np.linalg.cond(st['L2'] if 'L2' in st and st['L2'] is not None else st['L1'])) ** 2      # of the factored matrix
    e['cond'] = cond
    # the mean carries cond * eps of solve error whatever computes it: 1e-10 up to cond ~ 2e5, the same multiple of cond beyond
    ok = e['mu'] <= max(1e-10, 5e-16 * cond) and e['var'] <= 1e-8 and e['ei'] <= 1e-8 and e['nll'] <= 1e-10 and (idx == int(np.argmax(ei_o)) or tie) \
        and mu_max == mu.max() and idx == int(np.argmax(ei))
    if not ok:
        bad += 1
        print('MISMATCH', dict(n=n, d=d, m=m, kernel=kernel, variant=variant, sY=sY, chunk=chunk), e, idx, int(np.argmax(ei_o)), flush=True)
    # likelihood + gradient of the same setting through the sweep entry point
    if d <= 16 and c % 3 == 0:
        nll, g, stat = gp.lml_batch(X, y, np.vstack([theta, theta * 1.1]), [sY, sY], kernel=kernel, grad=True)
        g_o = o.mle_grad(X, y, theta, sY, kernel=kernel)
        if stat[0] != 0 or abs(nll[0] - o.mlefcn(X, y, theta, sY, kernel)) > 1e-10 * abs(nll[0]) or np.max(np.abs(g[0] - g_o)) > 1e-7 * max(1.0, np.max(np.abs(g_o))):
            bad += 1
            print('LML MISMATCH', dict(n=n, d=d, kernel=kernel, sY=sY), nll[0], g[0], g_o, flush=True)
    if c % 10 == 9:
        print('[%d/%d] %.0f s worst so far %s' % (c + 1, cases, time.time() - t0, {k: '%.1e' % v for k, v in worst.items()}), flush=True)
print('cases %d  mismatches %d  worst %s' % (cases, bad, {k: '%.2e' % v for k, v in worst.items()}))
sys.exit(1 if bad else 0)
