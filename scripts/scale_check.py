"""Scale experiments (not a test): cfg3 shape on one GPU with a candidate slice, cfg5 and cfg4 timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP, synth
which = sys.argv[1:] or ['cfg3', 'cfg5', 'cfg4']
gp = GP(0)
if 'cfg3' in which:
    cfg = dict(synth.CONFIGS['cfg3'])
    m = 1 << 17
    t0 = time.time(); X, y, theta = synth.training_set(cfg); Xs = synth.candidates(cfg, 0, m); print('gen %.1fs' % (time.time() - t0), flush=True)
    t0 = time.time(); gp.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel']); t1 = time.time()
    tm = gp.timings()
    print('cfg3 fit n=32768 wall %.2fs' % (t1 - t0), {k: round(v, 1) for k, v in tm.items() if v and 'ms' in k}, 'chol TFLOP/s %.1f' % (32768 ** 3 / 3 / (tm['chol_ms'] / 2 * 1e-3) / 1e12), flush=True)
    gp.set_candidates(Xs)
    for it in range(2):
        t0 = time.time(); gp.posterior(); idx, best = gp.acquire(); t1 = time.time()
        tm = gp.timings()
        print('cfg3 sweep m=%d: wall %.3fs cand/s %.0f trmm TFLOP/s %.2f ks %.1f ms trmm %.1f ms' % (m, t1 - t0, m / (t1 - t0), tm['trmm_flops'] / tm['trmm_ms'] / 1e9, tm['ksbuild_ms'], tm['trmm_ms']), flush=True)
    # identities at training inputs: mu_i = y_i - sigma^2 alpha1_i
    gp.set_candidates(X[:4096]); gp.posterior(); mu, var = gp.read()
    _, al = gp.factor(want_L=False)
    a1 = al * gp.theta1
    print('identity mu: max|mu - (y - s2 a1)| / max|y| = %.2e' % (np.max(np.abs(mu - (y[:4096] - cfg['sigmaY'] ** 2 * a1[:4096]))) / np.max(np.abs(y))), 'var range', var.min(), var.max(), 'theta1', gp.theta1, flush=True)
if 'cfg5' in which:
    n, d, B = 16384, 8, 16
    X = synth.uniform(5, n * d).reshape(n, d); y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(105, n)
    thetas = np.outer(np.geomspace(0.25, 2.0, B), np.full(d, 0.5)); sig = np.geomspace(1e-2, 1, B)
    for grad in (False, True):
        t0 = time.time(); out = gp.lml_batch(X, y, thetas, sig, grad=grad); t1 = time.time()
        tm = gp.timings()
        flops = B * n ** 3 * (1.0 if grad else 1 / 3)
        print('cfg5 n=16384 B=%d grad=%s: %.2fs per setting %.3fs  %.1f TFLOP/s' % (B, grad, t1 - t0, (t1 - t0) / B, flops / (t1 - t0) / 1e12), {k: round(v, 1) for k, v in tm.items() if v and k in ('chol_ms', 'inverse_ms', 'trsv_ms', 'kbuild_ms', 'fit_total_ms')}, out[0], out[-1], flush=True)
if 'cfg4' in which:
    d, side = 4, 64
    g1 = (np.arange(side) + 0.5) / side
    grid = np.stack(np.meshgrid(g1, g1, g1, g1, indexing='ij'), -1).reshape(-1, d)
    f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
    n0, iters = 16, 30
    X0 = synth.uniform(4, n0 * d).reshape(n0, d)
    gi = GP(0)
    gi.reserve_appends(200)
    gi.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
    gi.set_candidates(grid)
    t0 = time.time(); gi.posterior(); print('cfg4 initial posterior m=%d: %.3fs' % (len(grid), time.time() - t0), flush=True)
    ts = []
    for it in range(iters):
        t0 = time.time()
        jj, ei = gi.acquire(bo.ACQ_EI, 0.01)
        gi.append(grid[jj], float(f(grid[jj])))
        ts.append(time.time() - t0)
        tm = gi.timings()
        if it % 10 == 9 or it < 2:
            nn = gi.n - 1
            print('cfg4 it %d n=%d: %.2f ms (sweep %.2f ms, acquire %.2f ms) -> %.0f GB/s on 8mn' % (it, nn, ts[-1] * 1e3, tm['append_sweep_ms'], tm['acquire_ms'], 8 * len(grid) * nn / (tm['append_sweep_ms'] * 1e-3) / 1e9), flush=True)
