"""GPU: candidates given as a lattice (the grid of the notebooks' nested loops, BayesOpt_v4.ipynb:705-714).  The points are
those of np.meshgrid; with the RBF kernel the frozen-theta loop updates the posterior through one product of per-axis kernel
factors per append instead of a pass over a resident (n, m) matrix -- checked against that pass (the same loop on the same
points given as a plain candidate array), against a refit on the enlarged training set, and against the CPU oracle."""
import numpy as np
import pytest

import bayesian_optimization_amd as bo
from conftest import TOL_MU, TOL_VAR, relmax
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def lattice(axes):
    return np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, len(axes))


def objective(x):
    return np.exp(-6 * np.sum((x - 0.4) ** 2, -1)) + 0.3 * np.sin(3 * x.sum(-1))


AXES = {1: [np.linspace(-1, 2, 41)],
        2: [np.linspace(0, 1, 9), np.linspace(0.1, 0.9, 130)],
        3: [np.linspace(0, 1, 7), np.linspace(0, 1, 5), np.linspace(-0.5, 0.5, 6)],
        4: [np.linspace(0, 1, 6), np.linspace(0, 1, 5), np.linspace(0, 1, 12), np.linspace(0, 1, 11)],
        5: [np.linspace(0, 1, 3), np.linspace(0, 1, 4), np.linspace(0, 1, 3), np.linspace(0, 1, 5), np.linspace(0, 1, 4)]}


@pytest.mark.parametrize('d', [1, 2, 3, 4, 5])
def test_lattice_points_are_the_meshgrid_in_lexicographic_order(gp0, d):
    axes = AXES[d]
    rng = np.random.default_rng(d)
    X = rng.random((20, d))
    gp0.fit(X, objective(X), np.full(d, 0.4), 0.05)
    gp0.set_candidates_lattice(axes)
    a = gp0.posterior()
    r1 = gp0.read()
    gp0.set_candidates(lattice(axes))
    b = gp0.posterior()
    r2 = gp0.read()
    assert a == b and np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1])
    with pytest.raises(ValueError):
        gp0.set_candidates_lattice(axes[:-1] if d > 1 else axes + axes)


@pytest.mark.parametrize('d', [1, 2, 3, 4, 5])
def test_lattice_loop_equals_the_streaming_loop_a_refit_and_the_oracle(d):
    """Twelve appends with theta frozen.  Lattice route and streaming route pick the same candidates and agree to rounding
    on every mean and variance; the end state equals a refit on the enlarged set and the CPU oracle's posterior."""
    from bayesian_optimization_amd import GP
    axes = AXES[d]
    grid = lattice(axes)
    rng = np.random.default_rng(10 + d)
    n0, steps = 9, 12
    X0 = rng.random((n0, d))
    theta = np.full(d, 0.35)
    runs = []
    for use_lattice in (True, False):
        gp = GP(0)
        gp.reserve_appends(steps)
        gp.fit(X0, objective(X0), theta, 1e-2, variant=bo.VARIANT_SINGLE)
        if use_lattice:
            gp.set_candidates_lattice(axes)
        else:
            gp.set_candidates(grid)
        mu_max = gp.posterior()
        j, v = gp.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        picks, Xa, ya = [j], [X0], [objective(X0)]
        for it in range(steps):
            if it % 2 == 0:
                mu_max, j, v = gp.append_acquire(grid[j], float(objective(grid[j])), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            else:
                mu_max = gp.append(grid[j], float(objective(grid[j])))
                j, v = gp.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
            Xa.append(grid[picks[-1]][None, :])
            ya.append(np.array([objective(grid[picks[-1]])]))
            picks.append(j)
            assert (gp.timings()['append_route'] >= 1.0) == (use_lattice and d > 1)      # one axis: nothing to factorise (2.0: the update launch carried the acquisition)
        mu, var = gp.read()
        assert mu_max == mu.max()
        runs.append(dict(picks=picks, mu=mu, var=var, theta1=gp.theta1, nll=gp.nll, X=np.vstack(Xa), y=np.concatenate(ya)))
        if use_lattice:
            gp.posterior()                                   # a full pass from the updated factor
            mu_f, var_f = gp.read()
            assert relmax(mu_f, mu) <= 1e-11 and relmax(var_f, var) <= 1e-10
            with pytest.raises(ValueError):
                gp.append(grid[0], 0.0)                      # the reserved rows are used up
        gp.close()
    a, b = runs
    assert a['picks'] == b['picks']
    assert relmax(a['mu'], b['mu']) <= 1e-12 and relmax(a['var'], b['var']) <= 1e-11
    assert a['theta1'] == b['theta1'] and a['nll'] == b['nll']          # the factor update is the same code on both routes
    ref = GP(0)
    ref.fit(a['X'], a['y'], theta, 1e-2, variant=bo.VARIANT_SINGLE)
    ref.set_candidates(grid)
    ref.posterior()
    mu_r, var_r = ref.read()
    ref.close()
    assert relmax(a['mu'], mu_r) <= TOL_MU and relmax(a['var'], var_r) <= TOL_VAR
    mu_o, var_o, _ = o.gprfcn_diag(grid, a['X'], a['y'], theta, 1e-2, variant=bo.VARIANT_SINGLE)
    assert relmax(a['mu'], mu_o) <= TOL_MU and relmax(a['var'], var_o) <= TOL_VAR


def test_matern_on_a_lattice_keeps_the_streaming_route():
    """Matern-5/2 of the Euclidean distance is not a product over the axes: the lattice only supplies the points."""
    from bayesian_optimization_amd import GP
    axes = AXES[3]
    grid = lattice(axes)
    X0 = np.random.default_rng(3).random((8, 3))
    gp = GP(0)
    gp.reserve_appends(3)
    gp.fit(X0, objective(X0), np.full(3, 0.5), 0.05, kernel=bo.KERNEL_MATERN52, variant=bo.VARIANT_SINGLE)
    gp.set_candidates_lattice(axes)
    gp.posterior()
    gp.append(grid[17], float(objective(grid[17])))
    assert gp.timings()['append_route'] == 0.0
    mu, var = gp.read()
    mu_o, var_o, _ = o.gprfcn_diag(grid, np.vstack([X0, grid[17][None, :]]), np.concatenate([objective(X0), [objective(grid[17])]]),
                                   np.full(3, 0.5), 0.05, kernel=bo.KERNEL_MATERN52, variant=bo.VARIANT_SINGLE)
    assert relmax(mu, mu_o) <= TOL_MU and relmax(var, var_o) <= TOL_VAR
    gp.close()


@pytest.mark.parametrize('contexts', [2, 3, 4])
def test_lattice_loop_over_several_contexts_is_bit_identical_to_one(contexts):
    """The lattice cut into slabs along axis 0 (6 points over 2, 3 or 4 contexts: slabs of different thickness): picks,
    acquisition values, max(Mu) and every mean / variance equal the one-context loop's bit for bit -- a candidate's sum runs
    over the training points in the same order wherever its tile sits."""
    from bayesian_optimization_amd import GP, MultiGP
    axes = AXES[4]
    grid = lattice(axes)
    X0 = np.random.default_rng(21).random((11, 4))
    theta = np.array([0.3, 0.4, 0.35, 0.5])
    out = []
    for gp in (GP(0), MultiGP([0] * contexts)):
        gp.reserve_appends(9)
        gp.fit(X0, objective(X0), theta, 1e-2, variant=bo.VARIANT_SINGLE)
        gp.set_candidates_lattice(axes)
        mu_max = gp.posterior()
        j, v = gp.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        seq = [(j, v, mu_max)]
        for _ in range(9):
            mu_max, j, v = gp.append_acquire(grid[j], float(objective(grid[j])), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            seq.append((j, v, mu_max))
        out.append((seq, gp.read(), gp.theta1, gp.nll))
        if isinstance(gp, MultiGP):
            assert all(g.timings()['append_route'] >= 1.0 for g in gp.contexts)
            assert [hi - lo for lo, hi in gp.blocks()] == [len(b) * (len(grid) // 6) for b in np.array_split(np.arange(6), contexts)]
        gp.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])
    assert out[0][2:] == out[1][2:]


def test_two_rank_lattice_loop_picks_the_same_points(gp0):
    """bench.py --config cfg4 --lattice on two ranks (slabs of axis 0, gloo rendezvous, both ranks on this GPU) against the
    one-process lattice loop: the same candidates, index for index."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from bayesian_optimization_amd import synth, dist as gdist
    side, iters, d, n0 = 16, 10, 4, 16
    env = dict(os.environ, GPBO_DIST_BACKEND='gloo', GPBO_FORCE_DEVICE='0', MASTER_ADDR='127.0.0.1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'cfg4', '--lattice', '--side', str(side),
           '--iters', str(iters), '--warmup', '0']
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    res = json.loads([ln for ln in proc.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['scaling'] == 'strong'
    g1 = (np.arange(side) + 0.5) / side
    f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
    X0 = synth.uniform(4, n0 * d).reshape(n0, d)
    gp0.reserve_appends(iters)
    gp0.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
    loop = gdist.ShardedLoop.on_lattice(gp0, [g1] * d)
    picked = []
    jj, _, x = loop.propose(bo.ACQ_EI, 0.01)
    for _ in range(iters):
        picked.append(int(jj))
        jj, _, x = loop.step(x, float(f(x)), bo.ACQ_EI, 0.01)
    gp0.reserve_appends(0)
    assert res['config']['picked_head'] == picked[:16]


def test_refit_with_other_length_scales_rebuilds_the_factor_tables():
    """Lattice candidates stay resident across refits (same column count); the per-axis factor tables belong to a fit and are
    rebuilt by the posterior that follows it: appends after a refit with other length scales, a larger training set and
    another reserve equal the oracle on the enlarged set."""
    from bayesian_optimization_amd import GP
    axes = AXES[4]
    grid = lattice(axes)
    rng = np.random.default_rng(77)
    gp = GP(0)
    Xa = rng.random((7, 4))
    for rnd, (theta, reserve, sY) in enumerate([(np.full(4, 0.3), 4, 1e-2), (np.array([0.5, 0.25, 0.4, 0.6]), 6, 0.05)]):
        gp.reserve_appends(reserve)
        gp.fit(Xa, objective(Xa), theta, sY, variant=bo.VARIANT_SINGLE)
        if rnd == 0:
            gp.set_candidates_lattice(axes)
        mu_max = gp.posterior()
        j, _ = gp.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
        for _ in range(reserve):
            Xa = np.vstack([Xa, grid[j][None, :]])
            mu_max, j, _ = gp.append_acquire(grid[j], float(objective(grid[j])), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
            assert gp.timings()['append_route'] >= 1.0
        mu, var = gp.read()
        mu_o, var_o, _ = o.gprfcn_diag(grid, Xa, objective(Xa), theta, sY, variant=bo.VARIANT_SINGLE)
        assert relmax(mu, mu_o) <= TOL_MU and relmax(var, var_o) <= TOL_VAR and mu_max == mu.max()
        ei_o = o.expected_improvement(mu_o.copy(), var_o, 0.01)
        assert j == int(np.argmax(ei_o))
    gp.close()


def test_one_launch_factor_update_equals_the_separate_launches():
    """Small models update the factor in one launch (chain of four phases in one workgroup, the lattice table row beside it);
    GPBO_APPEND_FUSED=0 keeps one launch per link.  The same loops in two child processes: picks, theta1, -log L, mean and
    variance bit for bit, on both routes, both kernels, and across the size limit of the one-launch form."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'check_fused_append.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'differing 0' in r.stdout


def test_acquisition_carried_by_the_update_launch_equals_the_separate_pass():
    """append_acquire on the lattice route evaluates the acquisition inside the update launch; with max(mu) as incumbent it
    speculates on where that maximum sits (the candidate that held it so far) and falls back to the separate pass when it
    moved.  GPBO_LATTICE_SPEC=0 keeps the separate pass.  The same loops in two child processes (EI with each incumbent
    mode, UCB, PI; plain appends in between; ragged lattices): picks, acquisition maxima, max(mu), theta1, -log L, mean and
    variance bit for bit, both outcomes of the speculation exercised, known incumbents always kept."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'check_lattice_spec.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'differing 0' in r.stdout
