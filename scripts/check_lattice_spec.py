#!/usr/bin/env python3
"""Lattice route, acquisition carried by the update launch (GPBO_LATTICE_SPEC=1, the default) against the acquisition in a
launch of its own (=0): the same loops in two child processes -- EI with the largest mean as incumbent (the case that
speculates on where that maximum sits and falls back when it moved), EI with the largest observation, EI with a given
incumbent, UCB and PI -- picks, acquisition maxima, max(mu), theta1, -log L and the final mean / variance arrays bit for bit;
and the carried run must have used both outcomes (kept / fallen back) in the speculative case.
    python3 scripts/check_lattice_spec.py"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out):
    sys.path.insert(0, ROOT)
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import GP
    rng = np.random.default_rng(9)
    res = {}
    cases = [(3, 20, 6, 60, bo.ACQ_EI, bo.INCUMBENT_MAX_MU), (4, 12, 9, 40, bo.ACQ_EI, bo.INCUMBENT_MAX_MU), (2, 70, 4, 50, bo.ACQ_EI, bo.INCUMBENT_MAX_Y),
             (3, 18, 8, 30, bo.ACQ_EI, bo.INCUMBENT_GIVEN), (3, 18, 8, 30, bo.ACQ_UCB, bo.INCUMBENT_MAX_MU), (2, 40, 5, 30, bo.ACQ_PI, bo.INCUMBENT_MAX_MU),
             (2, 33, 3, 25, bo.ACQ_EI, bo.INCUMBENT_MAX_MU)]
    for case, (d, side, n0, iters, acq, mode) in enumerate(cases):
        axes = [np.sort(rng.random(side + t)) for t in range(d)]          # ragged: partial tiles in both directions
        grid = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
        f = lambda x: float(np.sin(5 * x.sum()) + 0.3 * np.cos(9 * x[0]) + 0.5 * np.exp(-30 * np.sum((x - 0.7) ** 2)))
        X = rng.random((n0, d)); y = np.array([f(x) for x in X])
        g = GP(0)
        g.reserve_appends(iters)
        g.fit(X, y, np.full(d, 0.3), 1e-2, variant=bo.VARIANT_SINGLE)
        g.set_candidates_lattice(axes)
        mu_max = g.posterior()
        j, _ = g.acquire(acq, 0.3 if acq == bo.ACQ_UCB else 0.01, bo.INCUMBENT_GIVEN, mu_max)
        picks, best, mm, th, nl, route = [j], [], [], [], [], []
        for it in range(iters):
            if it % 11 == 10:          # a plain append in between: the next carried step has no location to speculate on
                m1 = g.append(grid[j], f(grid[j]))
                j, b = g.acquire(acq, 0.3 if acq == bo.ACQ_UCB else 0.01, mode, 0.25)
                route.append(-1.0)
            else:
                m1, j, b = g.append_acquire(grid[j], f(grid[j]), acq, 0.3 if acq == bo.ACQ_UCB else 0.01, mode, 0.25)
                route.append(g.timings()['append_route'])
            picks.append(j); best.append(b); mm.append(m1); th.append(g.theta1); nl.append(g.nll)
        mu, var = g.read()
        for k, v in (('picks', picks), ('best', best), ('mumax', mm), ('th', th), ('nl', nl), ('mu', mu), ('var', var)):
            res['%s%d' % (k, case)] = np.asarray(v)
        res['route%d' % case] = np.asarray(route)
        g.close()
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--child':
        child(sys.argv[2])
        sys.exit(0)
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for v in ('1', '0'):
            out = os.path.join(tmp, 'spec%s.npz' % v)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', out], check=True, timeout=400,
                           env=dict(os.environ, GPBO_LATTICE_SPEC=v))
            outs.append(dict(np.load(out)))
    bad = 0
    for k in sorted(outs[0]):
        if k.startswith('route'):
            continue
        same = outs[0][k].tobytes() == outs[1][k].tobytes()
        bad += not same
        if not same:
            print('DIFFERS', k, np.max(np.abs(outs[0][k].astype(float) - outs[1][k].astype(float))))
    r0 = outs[0]['route0']
    kept, fell = int(np.sum(r0 == 2.0)), int(np.sum(r0 == 1.0))
    print('speculative case: kept %d, fallen back %d; separate-launch run used the carried form %d times'
          % (kept, fell, sum(int(np.sum(outs[1][k] == 2.0)) for k in outs[1] if k.startswith('route'))))
    print('arrays compared %d  differing %d' % (len([k for k in outs[0] if not k.startswith('route')]), bad))
    ok = bad == 0 and kept > 0 and fell > 0 and all(np.all(outs[1][k] != 2.0) for k in outs[1] if k.startswith('route'))
    ok = ok and all(np.all(outs[0]['route%d' % c][outs[0]['route%d' % c] >= 0] == 2.0) for c in (2, 3, 4))   # known incumbent: always kept
    sys.exit(0 if ok else 1)
