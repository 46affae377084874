#!/usr/bin/env python3
"""The one-launch factor update of an append (small models) against the separate launches: the same loop in two child
processes, GPBO_APPEND_FUSED=1 and =0, on both routes (lattice and streaming), both kernels; picks, theta1, -log L and
the mean / variance arrays must agree bit for bit (the last two cases start just below the size limit of the one-launch
form and cross it inside the loop).
    python3 scripts/check_fused_append.py"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out):
    sys.path.insert(0, ROOT)
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import GP
    rng = np.random.default_rng(4)
    res = {}
    for case, (d, side, n0, iters, kernel, lattice) in enumerate([(2, 48, 5, 40, 0, True), (3, 20, 16, 60, 0, True), (4, 12, 9, 30, 0, True),
                                                                 (3, 20, 16, 60, 0, False), (2, 30, 7, 45, 1, False), (1, 700, 3, 25, 0, True),
                                                                 (2, 40, 505, 12, 0, True), (3, 16, 508, 8, 1, False)]):
        axes = [np.sort(rng.random(side)) for _ in range(d)]
        grid = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
        f = lambda x: float(np.sin(5 * x.sum()) + 0.3 * np.cos(9 * x[0]))
        X = rng.random((n0, d)); y = np.array([f(x) for x in X])
        g = GP(0)
        g.reserve_appends(iters)
        g.fit(X, y, np.full(d, 0.3), 1e-2, kernel=kernel, variant=bo.VARIANT_SINGLE)
        if lattice:
            g.set_candidates_lattice(axes)
        else:
            g.set_candidates(grid)
        mu_max = g.posterior()
        j, _ = g.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        picks, th, nl = [j], [], []
        for _ in range(iters):
            mu_max, j, best = g.append_acquire(grid[j], f(grid[j]), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            picks.append(j); th.append(g.theta1); nl.append(g.nll)
        mu, var = g.read()
        res['picks%d' % case] = np.array(picks); res['th%d' % case] = np.array(th); res['nl%d' % case] = np.array(nl)
        res['mu%d' % case] = mu; res['var%d' % case] = var
        g.close()
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--child':
        child(sys.argv[2])
        sys.exit(0)
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for v in ('1', '0'):
            out = os.path.join(tmp, 'fused%s.npz' % v)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', out], check=True, timeout=300,
                           env=dict(os.environ, GPBO_APPEND_FUSED=v))
            outs.append(dict(np.load(out)))
    bad = 0
    for k in sorted(outs[0]):
        same = outs[0][k].tobytes() == outs[1][k].tobytes()
        bad += not same
        if not same:
            print('DIFFERS', k, np.max(np.abs(outs[0][k].astype(float) - outs[1][k].astype(float))))
    print('arrays compared %d  differing %d' % (len(outs[0]), bad))
    sys.exit(1 if bad else 0)
