#!/usr/bin/env python3
"""Randomised differential run of the several-contexts handle against one context: random shapes (candidate counts below,
at and above the number of contexts, training sizes around the block boundaries), both kernels, every posterior variant, the
trend flavour, appends, uploaded posteriors and NaN-carrying inputs.  Every number a split sweep returns -- mean, variance,
acquisition values, the pick, max(Mu), the model scalars -- must equal the one-context result BIT FOR BIT.  All contexts sit
on device 0, so one GPU is enough.      python3 scripts/fuzz_multi.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP, MultiGP

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
one = GP(0)
multis = {k: MultiGP([0] * k) for k in (2, 3, 4)}
bad = 0
t0 = time.time()


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def report(what, cfg):
    global bad
    bad += 1
    print('MISMATCH', what, cfg, flush=True)


for c in range(cases):
    k = int(rng.choice([2, 3, 4]))
    mg = multis[k]
    n = int(rng.choice([1, 2, 17, 127, 128, 129, 256, 300, 513]))
    d = int(rng.integers(1, 9))
    m = int(rng.choice([1, 2, 3, 5, 63, 257, 1000, 4097]))
    kernel = int(rng.integers(0, 2))
    sY = float(rng.choice([0.05, 0.1, 0.3]))
    theta = rng.uniform(0.3, 0.9, d) * (1 + 0.15 * d)
    X = rng.random((n, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n); Xs = rng.random((m, d))
    chunk = int(rng.choice([0, 256, 1024]))
    acq = int(rng.choice([bo.ACQ_EI, bo.ACQ_UCB, bo.ACQ_PI]))
    mode = int(rng.integers(0, 4))
    cfg = dict(case=c, k=k, n=n, d=d, m=m, kernel=kernel, sY=sY, chunk=chunk, acq=acq, mode=mode)
    for g in (one, mg):
        g.set_chunk(chunk)
    if mode == 3 and d <= 3:                                        # trend flavour
        deg = int(rng.integers(0, 3))
        th = float(theta[0])
        try:
            r1 = one.kriging_fit(X, y, th, deg)
        except Exception as e:
            r1 = type(e)
        try:
            r2 = mg.kriging_fit(X, y, th, deg)
        except Exception as e:
            r2 = type(e)
        if isinstance(r1, type) or isinstance(r2, type):
            if r1 is not r2:
                report('kriging fit outcome', cfg)
            continue
        if not (same(r1[0], r2[0]) and r1[1] == r2[1]):
            report('kriging beta/cov', cfg)
    else:
        variant = int(rng.choice([bo.VARIANT_V4, bo.VARIANT_SINGLE, bo.VARIANT_M3]))
        cfg['variant'] = variant
        if mode == 2 and n > 1:
            X = X.copy(); X[int(rng.integers(0, n)), 0] = np.nan   # a fit that must fail the same way on both
        try:
            one.fit(X, y, theta, sY, kernel=kernel, variant=variant); f1 = None
        except Exception as e:
            f1 = type(e)
        try:
            mg.fit(X, y, theta, sY, kernel=kernel, variant=variant); f2 = None
        except Exception as e:
            f2 = type(e)
        if f1 is not f2:
            report('fit outcome %s vs %s' % (f1, f2), cfg)
        if f1 is not None or f2 is not None:
            continue
        if not (one.theta1 == mg.theta1 and one.nll == mg.nll and one.n == mg.n):
            report('model scalars', cfg)
    if mode == 1 and m > 2:
        Xs = Xs.copy(); Xs[int(rng.integers(0, m)), 0] = np.nan   # one NaN candidate: max(Mu) and the pick follow NumPy
    one.set_candidates(Xs); mg.set_candidates(Xs)
    a, b = one.posterior(), mg.posterior()
    if not same(a, b):
        report('max(Mu) %r vs %r' % (a, b), cfg)
    inc_mode = int(rng.choice([bo.INCUMBENT_MAX_MU, bo.INCUMBENT_MAX_Y]))
    p1, p2 = one.acquire(acq, 0.01, inc_mode), mg.acquire(acq, 0.01, inc_mode)
    if not (p1[0] == p2[0] and same(p1[1], p2[1])):
        report('pick %r vs %r' % (p1, p2), cfg)
    r1, r2 = one.read(acq=True), mg.read(acq=True)
    if not all(same(u, v) for u, v in zip(r1, r2)):
        report('mean / variance / acquisition arrays', cfg)
    if m > 7:
        off, cnt = int(rng.integers(0, m - 3)), 3
        if not all(same(u, v) for u, v in zip(one.read(offset=off, count=cnt), mg.read(offset=off, count=cnt))):
            report('ranged read', cfg)
    if np.all(np.isfinite(r1[2])):
        thr = float(np.median(r1[2]))
        if one.first_at_least(thr) != mg.first_at_least(thr):
            report('first_at_least', cfg)
    if c % 10 == 9:
        print('[%d/%d] %.0f s, mismatches %d' % (c + 1, cases, time.time() - t0, bad), flush=True)

# the frozen-theta loop: replicated model, split candidates -- random points (streaming update) or a random lattice cut
# into slabs along its first axis (update by a product of per-axis kernel factors)
for c in range(max(cases // 5, 4)):
    k = int(rng.choice([2, 3, 4]))
    mg = multis[k]
    d = int(rng.integers(1, 5)); n0 = int(rng.integers(2, 40)); m = int(rng.choice([50, 333, 2000])); steps = int(rng.integers(1, 12))
    X = rng.random((n0, d)); f = lambda x: np.exp(-6 * np.sum((x - 0.4) ** 2, -1)); Xs = rng.random((m, d))
    axes = None
    if c % 2 == 1:
        axes = [np.sort(rng.random(int(rng.integers(k, 12)))) for _ in range(d)]      # at least one axis-0 point per context
        Xs = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
        m = len(Xs)
    theta = np.full(d, 0.3)
    picks = []
    for g in (one, mg):
        g.set_chunk(0)
        g.reserve_appends(steps)
        g.fit(X, f(X), theta, 1e-2, variant=bo.VARIANT_SINGLE)
        if axes is None:
            g.set_candidates(Xs)
        else:
            g.set_candidates_lattice(axes)
        mu_max = g.posterior()
        j, v = g.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        seq = [(j, v, mu_max)]
        for _ in range(steps):
            mu_max, j, v = g.append_acquire(Xs[j], float(f(Xs[j])), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            seq.append((j, v, mu_max))
        picks.append((seq, g.read(), g.theta1, g.nll))
        g.reserve_appends(0)
    if not (picks[0][0] == picks[1][0] and all(same(u, v) for u, v in zip(picks[0][1], picks[1][1])) and picks[0][2:] == picks[1][2:]):
        report('append loop', dict(k=k, d=d, n0=n0, m=m, steps=steps, lattice=axes is not None))

print('cases %d  mismatches %d  (%.0f s)' % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
