#!/usr/bin/env python3
"""Randomised call sequences on ONE context against a shadow model kept with the CPU oracle: fits of every variant, kriging
fits, candidate uploads, posteriors, acquisitions, appends, uploaded posteriors and likelihood sweeps in any order, including
the orders that are misuse.  A call must either return what the oracle says for the state the shadow holds, or refuse with
ValueError / LinAlgError where the shadow says the call has nothing to work on; it must never return numbers that belong
to an earlier model or candidate set.  python3 scripts/fuzz_state_machine.py [sequences] [steps] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP
from oracle import gp_oracle as o

seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))
REFUSE = (ValueError, np.linalg.LinAlgError)
bad = 0
counts = {}
t0 = time.time()


def report(seq, step, op, what):
    global bad
    bad += 1
    print('MISMATCH seq %d step %d %s: %s' % (seq, step, op, what), flush=True)


def oracle_posterior(S):
    if S['kind'] == 'gp':
        mu, var, _ = o.gprfcn_diag(S['Xs'], S['X'], S['y'], S['theta'], S['sY'], kernel=S['kernel'], variant=S['variant'])
        return mu, var
    kst = o.kriging_fit(S['X'], S['y'], S['ktheta'], S['deg'])
    return o.kriging_predict(kst, S['Xs'])


for seq in range(seqs):
    gp = GP(0)
    S = dict(kind=None, d=0, Xs=None, post=None, reserve=int(rng.choice([0, 4])), left=0, inc=False, acq=False)
    gp.reserve_appends(S['reserve'])
    for step in range(steps):
        op = str(rng.choice(['fit', 'fit', 'kriging', 'cands', 'cands', 'posterior', 'posterior', 'posterior', 'acquire', 'append', 'append', 'append',
                             'upload', 'lml', 'reserve', 'read', 'readacq', 'chunk', 'predict']))
        counts[op] = counts.get(op, 0) + 1
        try:
            if op == 'fit':
                n, d = int(rng.choice([1, 5, 40, 127, 129, 260])), int(rng.integers(1, 5))
                if S['d'] and rng.random() < 0.6:
                    d = S['d']
                X = rng.random((n, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n)
                theta = rng.uniform(0.3, 0.9, d) * (1 + 0.15 * d)
                sY, kernel = float(rng.choice([0.05, 0.2])), int(rng.integers(0, 2))
                variant = int(rng.choice([bo.VARIANT_V4, bo.VARIANT_SINGLE, bo.VARIANT_SINGLE, bo.VARIANT_M3]))
                gp.fit(X, y, theta, sY, kernel=kernel, variant=variant)
                st = o.fit(X, y, theta, sY, kernel=kernel, variant=variant)
                if abs(gp.theta1 - st['theta1']) > 1e-10 * st['theta1'] or abs(gp.nll - st['nll']) > 1e-10 * max(abs(st['nll']), 1.0):
                    report(seq, step, op, 'theta1/nll %r %r vs %r %r' % (gp.theta1, gp.nll, st['theta1'], st['nll']))
                if d != S['d']:
                    S['Xs'] = None
                S.update(kind='gp', d=d, X=X, y=y, theta=theta, sY=sY, kernel=kernel, variant=variant, post=None, inc=False, acq=False,
                         left=S['reserve'] if variant == bo.VARIANT_SINGLE else 0)
            elif op == 'kriging':
                n, d, deg = int(rng.choice([12, 40, 90])), int(rng.integers(1, 4)), int(rng.integers(0, 3))
                if S['d'] and S['d'] <= 3 and rng.random() < 0.6:
                    d = S['d']
                X = rng.random((n, d)) * 4.0; y = np.sin(X.sum(1)) + 0.05 * rng.standard_normal(n)
                kth = float(rng.uniform(0.08, 0.2)) * (1.0 if d == 1 else 2.0)
                kst = o.kriging_fit(X, y, kth, deg)
                if np.linalg.cond(kst['K']) > 1e6:
                    continue
                beta, cov = gp.kriging_fit(X, y, kth, deg)
                if rel(beta, kst['beta']) > 1e-8 or abs(cov - kst['cov']) > 1e-8 * abs(kst['cov']):
                    report(seq, step, op, 'beta/cov')
                if d != S['d']:
                    S['Xs'] = None
                S.update(kind='kriging', d=d, X=X, y=y, ktheta=kth, deg=deg, post=None, inc=False, acq=False, left=0)
            elif op == 'cands':
                m = int(rng.choice([1, 3, 255, 257, 1500]))
                d = S['d'] if S['d'] else 2
                Xs = rng.random((m, d)) * (4.0 if S['kind'] == 'kriging' else 1.0)
                if S['kind'] is None:
                    try:
                        gp.set_candidates(Xs)
                        report(seq, step, op, 'accepted candidates without a model')
                    except REFUSE:
                        pass
                    continue
                if rng.random() < 0.4:      # the same kind of set handed over as a lattice (per-axis coordinates)
                    axes = [np.sort(rng.random(int(rng.integers(1, 9)))) * (4.0 if S['kind'] == 'kriging' else 1.0) for _ in range(d)]
                    gp.set_candidates_lattice(axes)
                    Xs = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d)
                    counts['lattice'] = counts.get('lattice', 0) + 1
                else:
                    gp.set_candidates(Xs)
                S.update(Xs=Xs, post=None, inc=False, acq=False)
            elif op == 'posterior':
                usable = S['kind'] is not None and S['Xs'] is not None      # (an uploaded posterior is simply replaced)
                if not usable:
                    try:
                        gp.posterior()
                        report(seq, step, op, 'returned without model or candidates (kind %r, candidates %r, posterior %r)'
                               % (S['kind'], S['Xs'] is not None, S['post']))
                    except REFUSE:
                        pass
                    continue
                mm = gp.posterior()
                counts['posterior carried out'] = counts.get('posterior carried out', 0) + 1
                mu_o, var_o = oracle_posterior(S)
                mu, var = gp.read()
                scale = max(float(np.max(np.abs(var_o))), 1e-300)
                if len(mu) != len(mu_o) or rel(mu, mu_o) > 1e-8 or float(np.max(np.abs(var - var_o))) > 1e-8 * scale or abs(mm - np.max(mu_o)) > 1e-8 * max(abs(np.max(mu_o)), 1e-12):
                    report(seq, step, op, 'mu %.2e var %.2e (kind %s)' % (rel(mu, mu_o), float(np.max(np.abs(var - var_o))) / scale, S['kind']))
                S.update(post='model', mu=mu_o, var=var_o, inc=(S['kind'] == 'gp' and S['variant'] == bo.VARIANT_SINGLE and S['reserve'] > 0), acq=False)
            elif op == 'acquire':
                if S['post'] is None:
                    try:
                        gp.acquire()
                        report(seq, step, op, 'returned without a posterior')
                    except REFUSE:
                        pass
                    continue
                kappa = float(rng.choice([0.0, 0.01]))
                idx, top = gp.acquire(bo.ACQ_EI, kappa, bo.INCUMBENT_MAX_MU)
                ref = np.atleast_1d(o.expected_improvement(S['mu'].copy(), np.maximum(S['var'], 0.0), kappa))
                vals = np.atleast_1d(gp.read(mu=False, var=False, acq=True))
                # negative mse of the kriging flavour gives NaN on both sides
                fin = np.isfinite(ref)
                if len(vals) != len(ref) or (fin.any() and float(np.max(np.abs(vals[fin] - ref[fin]))) > 1e-7 * max(float(np.max(np.abs(ref[fin]))), 1e-300) + 1e-12):
                    report(seq, step, op, 'EI values')
                elif np.isfinite(vals).any() and (idx != int(np.nanargmax(vals)) or top != vals[idx]):
                    report(seq, step, op, 'argmax %d vs own first maximum %d' % (idx, int(np.nanargmax(vals))))
                S['acq'] = True
                S['acqvals'] = vals.copy()
            elif op == 'append':
                can = S['kind'] == 'gp' and S['variant'] == bo.VARIANT_SINGLE and S['inc'] and S['left'] > 0 and S['post'] == 'model'
                x_new = rng.random(S['d'] if S['d'] else 2); y_new = float(rng.standard_normal())
                if not can:
                    try:
                        gp.append(x_new, y_new)
                        report(seq, step, op, 'accepted (kind %r variant %r incremental %r left %d posterior %r)'
                               % (S['kind'], S.get('variant'), S['inc'], S['left'], S['post']))
                    except REFUSE:
                        pass
                    continue
                fusedcall = rng.random() < 0.5          # append + acquisition in one call (the routes may carry it in the update launch)
                if fusedcall:
                    acqk, modek = int(rng.integers(0, 3)), int(rng.integers(0, 3))
                    mm1, jj, bb = gp.append_acquire(x_new, y_new, acqk, 0.05, modek, 0.1)
                else:
                    gp.append(x_new, y_new)
                counts['append carried out'] = counts.get('append carried out', 0) + 1
                S['X'] = np.vstack([S['X'], x_new[None, :]]); S['y'] = np.append(S['y'], y_new); S['left'] -= 1
                mu_o, var_o = oracle_posterior(S)
                mu, var = gp.read()
                if fusedcall:
                    # against the two-call form on the device's own posterior: same maximum, same first index, same max(mu)
                    inc = {0: float(mu.max()), 1: float(S['y'].max()), 2: 0.1}[modek]
                    sg = np.sqrt(var)
                    with np.errstate(all='ignore'):
                        if acqk == 1:
                            vals = mu + 0.05 * sg
                        else:
                            vals = o.expected_improvement(mu.copy(), var, 0.05, incumbent=inc) if acqk == 0 else o.probability_of_improvement(mu, inc, sg)
                    if mm1 != mu.max():
                        report(seq, step, op, 'append_acquire max(mu) %r vs %r' % (mm1, float(mu.max())))
                    if vals is not None and np.all(np.isfinite(vals)):
                        top = float(vals.max())
                        if abs(bb - top) > 1e-9 * max(abs(top), 1e-300) + 1e-300 or vals[jj] < top - 1e-9 * max(abs(top), 1e-300):
                            report(seq, step, op, 'append_acquire acq %d mode %d: best %r at %d, own maximum %r at %d' % (acqk, modek, bb, jj, top, int(np.argmax(vals))))
                if rel(mu, mu_o) > 1e-8 or float(np.max(np.abs(var - var_o))) > 1e-8 * max(float(np.max(np.abs(var_o))), 1e-300) or gp.n != len(S['y']):
                    report(seq, step, op, 'mu %.2e var %.2e n %d' % (rel(mu, mu_o), float(np.max(np.abs(var - var_o))), gp.n))
                S.update(mu=mu_o, var=var_o, acq=False)
            elif op == 'upload':
                m = int(rng.choice([1, 100, 3000]))
                mu, var = rng.standard_normal(m), rng.random(m)
                gp.set_posterior(mu, var)
                a, b = gp.read()
                if not (np.array_equal(a, mu) and np.array_equal(b, var)):
                    report(seq, step, op, 'read back differs')
                S.update(post='uploaded', mu=mu, var=var, inc=False, acq=False)
            elif op == 'lml':
                n, d, B = int(rng.choice([3, 60, 130, 300])), int(rng.integers(1, 5)), int(rng.choice([1, 1, 4]))
                X = rng.random((n, d)); y = np.sin(3 * X.sum(1)) + 0.1 * rng.standard_normal(n)
                th = rng.uniform(0.3, 0.9, (B, d)) * (1 + 0.15 * d); sg = rng.choice([0.05, 0.3], B)
                nll, stt = gp.lml_batch(X, y, th, sg)
                for b in range(B):
                    ref = o.mlefcn(X, y, th[b], float(sg[b]))
                    if stt[b] != 0 or abs(nll[b] - ref) > 1e-10 * max(abs(ref), 1.0):
                        report(seq, step, op, 'nll %r vs %r' % (nll[b], ref))
                # the sweep runs in private work contexts: model, candidates and posterior of this context are as before
            elif op == 'reserve':
                S['reserve'] = int(rng.choice([0, 1, 3, 10]))
                gp.reserve_appends(S['reserve'])          # takes effect at the next single-factor fit
            elif op == 'read':
                if S['post'] is None:
                    try:
                        gp.read()
                        if gp.m > 0:
                            report(seq, step, op, 'returned %d values without a posterior' % gp.m)
                    except REFUSE:
                        pass
                    continue
                mu, var = gp.read()
                if len(mu) != len(S['mu']) or rel(mu, S['mu']) > 1e-8:
                    report(seq, step, op, 'stale or wrong posterior: %d values, expected %d' % (len(mu), len(S['mu'])))
            elif op == 'readacq':
                if S['post'] is not None and S['acq']:
                    vals = np.atleast_1d(gp.read(mu=False, var=False, acq=True))
                    if len(vals) != len(S['mu']) or not np.array_equal(vals, S['acqvals'], equal_nan=True):
                        report(seq, step, op, 'stored acquisition values changed')
                else:
                    try:
                        gp.read(mu=False, var=False, acq=True)
                        if gp.m > 0:
                            report(seq, step, op, 'returned acquisition values that belong to no current posterior')
                    except REFUSE:
                        pass
                    try:
                        gp.first_at_least(0.0)
                        report(seq, step, op, 'first_at_least worked on stale acquisition values')
                    except REFUSE:
                        pass
            elif op == 'chunk':
                gp.set_chunk(int(rng.choice([0, 256, 1024])))
            elif op == 'predict':
                if S['kind'] != 'gp':
                    continue
                m = int(rng.choice([1, 300, 2000]))
                Xs = rng.random((m, S['d']))
                mu, var, ei, idx, eim, mum = gp.predict_ei(Xs, 0.01)
                S.update(Xs=Xs)
                mu_o, var_o = oracle_posterior(S)
                if rel(mu, mu_o) > 1e-8 or float(np.max(np.abs(var - var_o))) > 1e-8 * max(float(np.max(np.abs(var_o))), 1e-300) or idx != int(np.argmax(ei)):
                    report(seq, step, op, 'mu %.2e' % rel(mu, mu_o))
                S.update(post='model', mu=mu_o, var=var_o, inc=(S['variant'] == bo.VARIANT_SINGLE and S['reserve'] > 0), acq=True, acqvals=ei.copy())
        except REFUSE as e:
            report(seq, step, op, 'refused a valid call: %s (state kind %r d %r candidates %r posterior %r)'
                   % (e, S['kind'], S['d'], None if S['Xs'] is None else S['Xs'].shape, S['post']))
            break
    gp.close()
print('sequences %d x %d steps  ops %s  mismatches %d  (%.0f s)' % (seqs, steps, counts, bad, time.time() - t0))
sys.exit(1 if bad else 0)
