#!/usr/bin/env python3
"""Where the host's share of a lattice-loop iteration goes: the loop of bench.py's cfg4 block with a clock around the objective
function, the library call (append + sweep + acquisition, one synchronisation) and the bookkeeping.  python3 scripts/cfg4_host_turn.py [iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP, synth
from bayesian_optimization_amd import dist as gdist

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d, side, n0 = 4, 64, 16
g1 = (np.arange(side) + 0.5) / side
ii = np.arange(side ** d)
grid = np.stack([g1[(ii // side ** (d - 1 - k)) % side] for k in range(d)], -1)
f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
X0 = synth.uniform(4, n0 * d).reshape(n0, d)
gp = GP(0)
gp.reserve_appends(iters + 1)
gp.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
loop = gdist.ShardedLoop(gp, grid, 0)
jj, _, x = loop.propose(bo.ACQ_EI, 0.01)
jj, _, x = loop.step(x, float(f(x)), bo.ACQ_EI, 0.01)
gp.set_timing(False)
t_f = t_call = 0.0
t0 = time.perf_counter()
for it in range(iters):
    a = time.perf_counter()
    y = float(f(x))
    b = time.perf_counter()
    jj, ei, x = loop.step(x, y, bo.ACQ_EI, 0.01)
    c = time.perf_counter()
    t_f += b - a
    t_call += c - b
total = time.perf_counter() - t0
print('per iteration: total %.4f ms  objective %.4f ms  loop.step (Python wrapper + library call) %.4f ms' % (total / iters * 1e3, t_f / iters * 1e3, t_call / iters * 1e3))
