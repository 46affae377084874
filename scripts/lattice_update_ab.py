#!/usr/bin/env python3
"""Lattice-route update launch per model size, two settings of the library's environment switches side by side: the 64^4
loop of cfg4 in two child processes, sweep time (HIP events) of every iteration, and whether the picks agree.
    python3 scripts/lattice_update_ab.py ITERS VAR=VALUE [VAR=VALUE ...]     (the assignments apply to the second column)"""
import os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(iters):
    sys.path.insert(0, ROOT)
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import GP
    d, side, n0 = 4, 64, 16
    g1 = (np.arange(side) + 0.5) / side
    f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
    rng = np.random.default_rng(4)
    X0 = rng.random((n0, d))
    g = GP(0)
    g.reserve_appends(iters)
    g.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
    g.set_candidates_lattice([g1] * d)
    mu_max = g.posterior()
    j, _ = g.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
    out, picks = [], []
    for it in range(iters):
        x = np.array([g1[(j // side ** (d - 1 - k)) % side] for k in range(d)])
        picks.append(int(j))
        _, j, _ = g.append_acquire(x, float(f(x)), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
        out.append(g.timings()['append_sweep_ms'])
    mu, var = g.read()
    print(json.dumps({'sweep_ms': out, 'picks': picks, 'mu_sum': float(mu.sum()), 'var_sum': float(var.sum()), 'nll': g.nll}))


if __name__ == '__main__':
    if len(sys.argv) >= 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]))
        sys.exit(0)
    iters = int(sys.argv[1])
    setb = dict(a.split('=', 1) for a in sys.argv[2:])
    res = {}
    for name, env in (('a', {}), ('b', setb)):
        e = dict(os.environ)
        for k in setb:
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(iters)], check=True, capture_output=True, text=True, timeout=400, env=e)
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    same = all(res['a'][k] == res['b'][k] for k in ('picks', 'mu_sum', 'var_sum', 'nll'))
    print('same picks, sums of mean and variance, -log L:', same)
    q, b = np.array(res['a']['sweep_ms']), np.array(res['b']['sweep_ms'])
    print('   n   default ms   %s ms' % ' '.join(sys.argv[2:]))
    for i in range(0, iters, 16):
        print('%4d   %9.4f   %9.4f' % (16 + i, q[i:i + 16].mean(), b[i:i + 16].mean()))
    print(' all   %9.4f   %9.4f' % (q.mean(), b.mean()))
