#!/usr/bin/env python3
"""Device-memory drift over many create / fit / sweep / append / destroy cycles of every kind of handle (one context, several
contexts, lattice loops, likelihood sweeps, the trend flavour): free memory after the cycles must return to where it was
after the first one (a leak of any per-call or per-context allocation shows as a steady drift).
    python3 scripts/leak_check.py [cycles]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP, MultiGP

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)


def one_cycle(c):
    n, d, m = int(rng.choice([50, 300, 1100])), int(rng.integers(2, 5)), int(rng.choice([1000, 20000]))
    X = rng.random((n, d)); y = np.sin(3 * X.sum(1)); Xs = rng.random((m, d)); th = np.full(d, 0.5)
    gp = GP(0)
    gp.fit(X, y, th, 0.1, kernel=c % 2)
    gp.predict_ei(Xs, 0.01)
    gp.lml_batch(X, y, np.tile(th, (4, 1)) * np.linspace(0.8, 1.2, 4)[:, None], np.full(4, 0.1), grad=True)
    if d <= 3:
        gp.kriging_fit(X[:40] * 4, y[:40], 0.3, 1)
        gp.set_candidates(Xs * 4); gp.posterior()
    gp.close()
    mg = MultiGP([0, 0, 0])
    mg.reserve_appends(5)
    mg.fit(X[:30], y[:30], th, 0.05, variant=bo.VARIANT_SINGLE)
    axes = [np.linspace(0, 1, 9)] * d
    mg.set_candidates_lattice(axes)
    mu_max = mg.posterior()
    j, _ = mg.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
    grid = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
    for _ in range(5):
        mu_max, j, _ = mg.append_acquire(grid[j], float(np.sin(3 * grid[j].sum())), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
    mg.read()
    mg.close()


one_cycle(0)
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info(0)
for c in range(1, cycles + 1):
    one_cycle(c)
    if c % 10 == 0:
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info(0)
        print('after %3d cycles: free memory changed by %+.1f MiB' % (c, (free - free0) / 2 ** 20), flush=True)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info(0)
drift = (free0 - free1) / 2 ** 20
print('drift over %d cycles: %.1f MiB' % (cycles, drift))
sys.exit(1 if drift > 64 else 0)
