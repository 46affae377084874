#!/usr/bin/env python3
"""Randomised differential run of the lattice route of the frozen-theta loop: random lattices (1 - 5 axes, 1 - 14 points per
axis, unequal spacing), random length scales, noise levels and numbers of appends.  After the loop the lattice route's mean and
variance must equal (to rounding) those of the streaming route on the same points and of the CPU oracle refitted on the enlarged
training set; the two routes must have picked the same candidates unless the two best values were within 1e-9 of each other;
theta1 and -logL must be bit-identical between the routes.      python3 scripts/fuzz_lattice.py [cases] [seed]"""
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
worst = dict(mu_routes=0.0, var_routes=0.0, mu_oracle=0.0, var_oracle=0.0)
bad = 0
t0 = time.time()
for c in range(cases):
    d = int(rng.integers(1, 6))
    while True:
        axes = [np.sort(rng.uniform(-0.2, 1.2, int(rng.integers(1, 15)))) for _ in range(d)]
        if np.prod([len(a) for a in axes]) <= 60000:
            break
    grid = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
    m = len(grid)
    n0 = int(rng.integers(1, 150))
    steps = int(rng.integers(1, 16))
    theta = rng.uniform(0.25, 0.8, d) * (1 + 0.1 * d)
    sY = float(rng.choice([1e-2, 0.05, 0.2]))
    X0 = rng.random((n0, d))
    f = lambda x: np.sin(3 * x.sum(-1)) + np.exp(-5 * np.sum((x - 0.5) ** 2, -1))
    runs = []
    for lat in (True, False):
        gp = GP(0)
        gp.reserve_appends(steps)
        gp.fit(X0, f(X0), theta, sY, variant=bo.VARIANT_SINGLE)
        gp.set_candidates_lattice(axes) if lat else gp.set_candidates(grid)
        mu_max = gp.posterior()
        j, v = gp.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        picks, ties, Xa = [j], False, [X0]
        for _ in range(steps):
            Xa.append(grid[j][None, :])
            mu_max, j, v = gp.append_acquire(grid[j], float(f(grid[j])), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            picks.append(j)
        sizes = [len(x) for x in axes]
        Q = int(np.prod(sizes[(d + 1) // 2:]))
        assert (gp.timings()['append_route'] >= 1.0) == bool(lat and Q >= 2)     # (one axis, or fast axes of a single point: nothing to factorise)
        mu, var = gp.read()
        runs.append(dict(picks=picks, mu=mu, var=var, th1=gp.theta1, nll=gp.nll, X=np.vstack(Xa)))
        gp.close()
    a, b = runs
    same_path = a['picks'] == b['picks']
    cfg = dict(case=c, d=d, sizes=[len(x) for x in axes], n0=n0, steps=steps, sY=sY)
    if same_path:
        e = dict(mu_routes=rel(a['mu'], b['mu']), var_routes=rel(a['var'], b['var']))
        Xf = a['X']
        mu_o, var_o, st = o.gprfcn_diag(grid, Xf, f(Xf), theta, sY, variant=bo.VARIANT_SINGLE)
        e.update(mu_oracle=rel(a['mu'], mu_o), var_oracle=rel(a['var'], var_o))
        for k in worst:
            worst[k] = max(worst[k], e[k])
        cond = float(np.linalg.cond(st['L1'])) ** 2
        ok = e['mu_routes'] <= 1e-11 and e['var_routes'] <= 1e-10 and e['mu_oracle'] <= max(1e-10, 5e-16 * cond) and e['var_oracle'] <= 1e-8 \
            and a['th1'] == b['th1'] and a['nll'] == b['nll']
        if not ok:
            bad += 1
            print('MISMATCH', cfg, e, a['th1'] == b['th1'], flush=True)
    else:
        # the routes may part ways only at a numerical tie of the acquisition maximum: compare up to the first difference
        k = next(i for i, (p, q) in enumerate(zip(a['picks'], b['picks'])) if p != q)
        print('paths differ at step', k, cfg, a['picks'][k], b['picks'][k], flush=True)
        bad += 1
    if c % 10 == 9:
        print('[%d/%d] %.0f s worst so far %s' % (c + 1, cases, time.time() - t0, {k: '%.1e' % v for k, v in worst.items()}), flush=True)
print('cases %d  mismatches %d  worst %s' % (cases, bad, {k: '%.2e' % v for k, v in worst.items()}))
sys.exit(1 if bad else 0)
