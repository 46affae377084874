"""GPU: context state that earlier calls leave behind must never be read stale (work buffers of the likelihood sweep,
a posterior uploaded without candidates, a model padded for appends handed to another context, device-generated candidates)."""
import os

import numpy as np
import pytest

import bayesian_optimization_amd as bo
from conftest import ROOT, TOL_MU, TOL_VAR, relmax
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def _data(n, d, seed):
    X = o.splitmix64_uniform(seed, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1)) + 0.1 * (o.splitmix64_uniform(seed + 1, n) - 0.5)
    return X, y


def test_likelihood_sweep_leaves_the_callers_model_alone_and_survives_reallocation():
    from bayesian_optimization_amd import GP
    gp = GP(0)
    Xa, ya = _data(200, 3, 5)
    Xb, yb = _data(450, 3, 9)
    th = np.array([[0.4, 0.5, 0.6]])
    ref_a = o.mlefcn(Xa, ya, th[0], 0.1)
    ref_b = o.mlefcn(Xb, yb, th[0], 0.1)
    for _ in range(2):                                # single setting, n > 128: the stream-lane path with graph replay
        nll, stat = gp.lml_batch(Xa, ya, th, [0.1])
        assert stat[0] == 0 and abs(nll[0] - ref_a) <= 1e-10 * abs(ref_a)
    # a fit of another size on the same context, then candidates and a posterior
    gp.fit(Xb, yb, th[0], 0.1)
    Xs = o.splitmix64_uniform(3, 900).reshape(300, 3)
    gp.set_candidates(Xs)
    gp.posterior()
    mu0, var0 = gp.read()
    # the same sweep again (another n in between as well): right answers, and the fitted model is untouched
    nll, stat = gp.lml_batch(Xb, yb, th, [0.1])
    assert stat[0] == 0 and abs(nll[0] - ref_b) <= 1e-10 * abs(ref_b)
    nll, g, stat = gp.lml_batch(Xa, ya, th, [0.1], grad=True)
    assert stat[0] == 0 and abs(nll[0] - ref_a) <= 1e-10 * abs(ref_a)
    assert np.max(np.abs(g[0] - o.mle_grad(Xa, ya, th[0], 0.1))) <= 1e-8 * np.max(np.abs(g[0]))
    assert gp.n == 450
    gp.posterior()
    mu1, var1 = gp.read()
    assert np.array_equal(mu0, mu1) and np.array_equal(var0, var1)
    mu_o, var_o, _ = o.gprfcn_diag(Xs, Xb, yb, th[0], 0.1)
    assert relmax(mu1, mu_o) <= TOL_MU and relmax(var1, var_o) <= TOL_VAR
    gp.close()


def test_uploaded_posterior_does_not_pose_as_candidates(gp0):
    """An uploaded (mu, var) pair of any length replaces the posterior arrays only: the resident candidates stay what they
    were, posterior() recomputes over exactly them (never over rows the longer upload would suggest), and a refit with the
    same column count keeps them too."""
    X, y = _data(150, 2, 21)
    gp0.fit(X, y, np.array([0.3, 0.4]), 0.05)
    Xs = o.splitmix64_uniform(8, 400).reshape(200, 2)
    gp0.set_candidates(Xs)
    gp0.posterior()
    mu, var = gp0.read()
    for big in (5000, 200, 7):                        # longer than, as long as, shorter than the resident candidate set
        gp0.set_posterior(np.linspace(0, 1, big), np.full(big, 0.25))
        idx, best = gp0.acquire(bo.ACQ_UCB, 1.0)
        assert idx == big - 1 and best == 1.0 + 0.5 and gp0.m == big
        gp0.posterior()
        mu2, var2 = gp0.read()
        assert gp0.m == 200 and np.array_equal(mu, mu2) and np.array_equal(var, var2)
    gp0.set_posterior(np.linspace(0, 1, 5000), np.full(5000, 0.25))
    gp0.fit(X, y, np.array([0.3, 0.4]), 0.05)         # the uploaded posterior goes with the old model, the candidates stay
    with pytest.raises(ValueError):
        gp0.read()
    gp0.posterior()
    mu3, var3 = gp0.read()
    assert np.array_equal(mu, mu3) and np.array_equal(var, var3)
    gp0.fit(np.hstack([X, X[:, :1]]), y, np.array([0.3, 0.4, 0.5]), 0.05)      # another column count: the candidates are dropped
    with pytest.raises(ValueError):
        gp0.posterior()


def test_model_padded_for_appends_hands_over_with_its_padding(gp0):
    """A single-factor model fitted after reserve_appends pads past round_up(n, 128); the receiver must size its buffers
    by the sender's n_pad (what broadcast_model sends)."""
    import torch
    from bayesian_optimization_amd import GP
    from bayesian_optimization_amd import dist as gdist
    X, y = _data(120, 3, 31)
    Xs = o.splitmix64_uniform(32, 3000).reshape(1000, 3)
    src = GP(0)
    src.reserve_appends(200)                          # n_pad = 384, not 128
    src.fit(X, y, np.array([0.5, 0.6, 0.7]), 0.1, variant=bo.VARIANT_SINGLE)
    ts, sd = gdist.model_tensors(src)
    assert sd.n_pad == 384
    dst = GP(0)
    dst.state_alloc(sd.n, sd.d, sd.kernel, sd.variant, sd.sigmaY, n_pad=sd.n_pad)
    td, sd2 = gdist.model_tensors(dst)
    assert (sd2.W_bytes, sd2.alpha_bytes, sd2.Xts_bytes) == (sd.W_bytes, sd.alpha_bytes, sd.Xts_bytes)
    for a, b in zip(td, ts):
        a.copy_(b)
    torch.cuda.synchronize()
    dst.state_commit(sd.y_max)
    r0 = src.predict_ei(Xs, 0.01)
    r1 = dst.predict_ei(Xs, 0.01)
    for u, v in zip(r0, r1):
        assert np.array_equal(u, v)
    with pytest.raises(ValueError):
        dst.state_alloc(sd.n, sd.d, sd.kernel, sd.variant, sd.sigmaY, n_pad=100)
    src.close()
    dst.close()


def test_device_generated_candidates_equal_the_host_stream(gp0):
    from bayesian_optimization_amd import synth
    cfg = dict(synth.CONFIGS['cfg2'], n=256)
    X, y, theta = synth.training_set(cfg)
    gp0.fit(X, y, theta, cfg['sigmaY'])
    lo, m = 12345, 7001
    gp0.set_candidates(synth.candidates(cfg, lo, lo + m))
    gp0.posterior()
    a = gp0.read()
    gp0.set_candidates_uniform(cfg['seed_c'], lo, m)
    gp0.posterior()
    b = gp0.read()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(synth.uniform(7, 50, offset=3), o.splitmix64_uniform(7, 50, offset=3))


def test_fused_loop_step_equals_append_then_acquire():
    """gpbo_append_acquire (one call, one synchronisation, acquisition array not stored) against gpbo_append + gpbo_acquire:
    the same picks, maxima and posterior, bit for bit, over a run of iterations; odd and even candidate counts (the sweep
    has a one- and a two-candidates-per-thread form)."""
    from bayesian_optimization_amd import GP
    from bayesian_optimization_amd import dist as gdist
    for m in (30001, 65536):
        X, y = _data(40, 3, 77)
        Xs = o.splitmix64_uniform(78, m * 3).reshape(m, 3)
        f = lambda x: float(np.sin(3 * x.sum()) + 0.2 * x[0])
        ga, gb = GP(0), GP(0)
        loops = []
        for gp in (ga, gb):
            gp.reserve_appends(12)
            gp.fit(X, y, np.array([0.4, 0.5, 0.6]), 0.05, variant=bo.VARIANT_SINGLE)
            loops.append(gdist.ShardedLoop(gp, Xs, 0))
        ja, _, xa = loops[0].propose(bo.ACQ_EI, 0.01)
        jb, _, xb = loops[1].propose(bo.ACQ_EI, 0.01)
        assert ja == jb
        for it in range(12):
            loops[0].observe(xa, f(xa))                               # two calls, two round trips
            ja, ea, xa = loops[0].propose(bo.ACQ_EI, 0.01)
            jb, eb, xb = loops[1].step(xb, f(xb), bo.ACQ_EI, 0.01)    # one fused call
            assert (ja, ea) == (jb, eb) and np.array_equal(xa, xb) and loops[0].mu_max == loops[1].mu_max
        ra, rb = ga.read(), gb.read()
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        assert ga.n == gb.n == 52 and ga.theta1 == gb.theta1 and ga.nll == gb.nll and ga.y_max == gb.y_max
        ga.close()
        gb.close()


def test_factorisation_kernel_choices_agree():
    """The factorisation picks its kernels by size (register-resident or tile-engine panel solve, block inverses inside the
    diagonal kernel or after the factorisation, whole, half or quarter tiles for the trailing update and the inverse).  Forced to
    the forms that large problems use, a mid-size fit must give the same model: theta1, -logL, factor and posterior."""
    import json
    import subprocess
    import sys
    code = r'''
import os, sys, json, numpy as np
sys.path.insert(0, os.getcwd())
from bayesian_optimization_amd import GP, synth
n, d = 700, 3
X = synth.uniform(31, n * d).reshape(n, d)
y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(32, n)
Xs = synth.uniform(33, 500 * d).reshape(500, d)
gp = GP(0)
gp.fit(X, y, np.array([0.4, 0.5, 0.6]), 0.05)
mu, var, ei, idx, _, _ = gp.predict_ei(Xs, 0.01)
L, a = gp.factor()
nll, g, st = gp.lml_batch(X, y, np.array([[0.4, 0.5, 0.6], [0.3, 0.3, 0.3]]), [0.05, 0.1], grad=True)
np.savez(sys.argv[1], mu=mu, var=var, L=L, a=a, th1=gp.theta1, nll=gp.nll, idx=idx, lml=nll, g=g)
'''
    outs = []
    noq = {'GPBO_SYRK_QUARTER_MAX': '0', 'GPBO_SYRK_RECT_QUARTER_MAX': '0', 'GPBO_INV_QUARTER_MAX': '0'}
    forms = [{},                                                                                   # the defaults (quarter tiles at this size)
             dict(noq, GPBO_TRSM_ROWS_MAX='0', GPBO_SYRK_HALF_MAX='0', GPBO_INV_HALF_MAX='0'),     # 128 x 128 tiles everywhere
             dict(noq, GPBO_CHOL_GROUP='4', GPBO_SYRK_HALF_MAX='100000', GPBO_INV_HALF_MAX='100000'),   # half tiles, groups of panels
             {'GPBO_CHOL_GROUP': '4', 'GPBO_SYRK_QUARTER_MAX': '100000', 'GPBO_SYRK_RECT_QUARTER_MAX': '100000',
              'GPBO_INV_QUARTER_MAX': '100000'}]                                                   # quarter tiles, groups of panels
    for k, env_add in enumerate(forms):
        path = os.path.join(ROOT, 'gpurun_out', 'forms_%d.npz' % k)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ, **env_add)
        proc = subprocess.run([sys.executable, '-c', code, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=150)
        assert proc.returncode == 0, proc.stderr[-2000:]
        outs.append(np.load(path))
    ref = outs[0]
    rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
    # the forms differ in rounding only (substitution against multiplication by an explicit inverse in the panel solve);
    # what that leaves scales with the conditioning (here ~1e4): three orders of magnitude below the parity targets
    for o2 in outs[1:]:
        assert abs(float(o2['th1']) - float(ref['th1'])) <= 1e-12 * float(ref['th1'])
        assert abs(float(o2['nll']) - float(ref['nll'])) <= 1e-12 * abs(float(ref['nll']))
        assert rel(o2['L'], ref['L']) <= 1e-11 and rel(o2['a'], ref['a']) <= 1e-10
        assert rel(o2['mu'], ref['mu']) <= 1e-10 and rel(o2['var'], ref['var']) <= 1e-10 and int(o2['idx']) == int(ref['idx'])
        assert rel(o2['lml'], ref['lml']) <= 1e-12 and rel(o2['g'], ref['g']) <= 1e-8


def test_tile_order_of_the_trailing_update_changes_no_bit(tmp_path):
    """Large trailing updates walk their tiles in bands of eight block rows instead of row by row (cache reuse at
    n = 32768).  Every tile is computed exactly once by the same arithmetic, so the factor must be bit-identical whatever
    the order: forced onto a 21-block-row problem (bands of 8 and 4 with a rectangle, a tip and an incomplete last band)
    and compared with the row-major order."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from bayesian_optimization_amd import GP, synth
n, d = 2600, 3
X = synth.uniform(41, n * d).reshape(n, d)
y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(42, n)
gp = GP(0)
gp.fit(X, y, np.array([0.4, 0.5, 0.6]), 0.05)
L, a = gp.factor()
import hashlib
np.savez(sys.argv[1], L=np.frombuffer(hashlib.sha256(np.ascontiguousarray(L).tobytes()).digest(), np.uint8), Lsum=np.abs(L).sum(),
         a=a, th1=gp.theta1, nll=gp.nll)
'''
    outs = []
    for k, band in enumerate(('0', '8', '4')):
        path = str(tmp_path / ('band_%d.npz' % k))
        env = dict(os.environ, GPBO_SYRK_BAND=band, GPBO_SYRK_HALF_MAX='0', GPBO_SYRK_QUARTER_MAX='0', GPBO_CHOL_GROUP='1')
        proc = subprocess.run([sys.executable, '-c', code, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=150)
        assert proc.returncode == 0, proc.stderr[-2000:]
        outs.append(np.load(path))
    assert np.isfinite(float(outs[0]['Lsum'])) and float(outs[0]['Lsum']) > 0 and float(outs[0]['th1']) > 0
    for o in outs[1:]:
        assert np.array_equal(o['L'], outs[0]['L']) and np.array_equal(o['a'], outs[0]['a'])
        assert float(o['th1']) == float(outs[0]['th1']) and float(o['nll']) == float(outs[0]['nll'])


def test_stored_acquisition_values_go_with_their_posterior(gp0):
    """gpbo_read(acq) and gpbo_first_at_least hand out what the last gpbo_acquire stored; once the posterior has changed
    (recomputed, uploaded, a point appended, append_acquire -- which stores nothing) they must refuse instead of
    returning the values of the earlier posterior."""
    X, y = _data(120, 2, 23)
    Xs = o.splitmix64_uniform(9, 600).reshape(300, 2)
    gp0.reserve_appends(2)
    gp0.fit(X, y, np.array([0.3, 0.4]), 0.05, variant=bo.VARIANT_SINGLE)
    gp0.set_candidates(Xs)
    gp0.posterior()
    with pytest.raises(ValueError):
        gp0.read(mu=False, var=False, acq=True)
    idx, best = gp0.acquire()
    ei = gp0.read(mu=False, var=False, acq=True)
    assert ei[idx] == best and gp0.first_at_least(best) == idx
    gp0.append(Xs[idx], 0.3)
    for call in (lambda: gp0.read(mu=False, var=False, acq=True), lambda: gp0.first_at_least(0.0)):
        with pytest.raises(ValueError):
            call()
    gp0.acquire()
    gp0.read(mu=False, var=False, acq=True)
    gp0.append_acquire(Xs[0], 0.1)
    with pytest.raises(ValueError):
        gp0.read(mu=False, var=False, acq=True)
    mu, var = gp0.read()                              # the posterior itself stays readable
    assert np.all(np.isfinite(mu)) and np.all(var >= 0)
    gp0.posterior()
    with pytest.raises(ValueError):
        gp0.first_at_least(0.0)
    gp0.reserve_appends(0)


def test_tile_order_of_the_candidate_sweep_and_small_kernel_width_change_no_bit(tmp_path):
    """GPBO_TRMM_COLGROUP only reorders the tiles of the variance product, and GPBO_SMALL_WIDE_MIN only picks the workgroup
    width of the one-workgroup likelihood kernel: every number is produced by the same arithmetic in the same order, so the
    posterior must be bit-identical whatever the tile order is; the two workgroup widths of the small kernel tile the matrix differently
    and must agree at least to round-off (30 points: the default takes the 256-thread form, the other settings the 1024-thread form)."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from bayesian_optimization_amd import GP, synth
n, d = 900, 3
X = synth.uniform(51, n * d).reshape(n, d)
y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(52, n)
Xs = synth.uniform(53, 5000 * d).reshape(5000, d)
gp = GP(0)
gp.fit(X, y, np.array([0.4, 0.5, 0.6]), 0.05)
mu, var, ei, idx, _, _ = gp.predict_ei(Xs, 0.01)
nll, g, st = gp.lml_batch(X[:30], y[:30], np.array([[0.4, 0.5, 0.6], [0.3, 0.3, 0.3]]), [0.05, 0.1], grad=True)
np.savez(sys.argv[1], mu=mu, var=var, ei=ei, idx=idx, lml=nll, g=g)
'''
    outs = []
    for k, env_add in enumerate(({}, {'GPBO_TRMM_COLGROUP': '3'}, {'GPBO_TRMM_COLGROUP': '1', 'GPBO_SMALL_WIDE_MIN': '16'},
                                 {'GPBO_SMALL_WIDE_MIN': '0'})):
        path = str(tmp_path / ('order_%d.npz' % k))
        proc = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, **env_add), cwd=ROOT, capture_output=True,
                              text=True, timeout=150)
        assert proc.returncode == 0, proc.stderr[-2000:]
        outs.append(np.load(path))
    for o2 in outs[1:]:
        for key in ('mu', 'var', 'ei', 'idx'):
            assert np.array_equal(o2[key], outs[0][key]), key
        assert np.all(np.isfinite(o2['lml'])) and relmax(o2['lml'], outs[0]['lml']) <= 1e-12 and relmax(o2['g'], outs[0]['g']) <= 1e-10


def test_nan_follows_numpy(gp0):
    """np.max(Mu) is NaN as soon as one mean is NaN (BayesOpt_v4.ipynb:193), and np.argmax(ei) returns the first NaN (:647):
    a poisoned posterior must surface, not be masked by the largest finite value."""
    m = 5000
    mu = np.sin(np.arange(m) * 0.01)
    var = np.full(m, 0.04)
    gp0.set_posterior(mu, var)
    idx, best = gp0.acquire(bo.ACQ_UCB, 1.0)
    u = mu + 1.0 * np.sqrt(var)
    assert idx == int(np.argmax(u)) and best == u[idx]
    mu2 = mu.copy()
    mu2[[4321, 777, 2500]] = np.nan
    gp0.set_posterior(mu2, var)
    idx, best = gp0.acquire(bo.ACQ_UCB, 1.0)
    assert idx == 777 and np.isnan(best) and idx == int(np.argmax(mu2 + np.sqrt(var)))
    # EI with the notebook's incumbent max(Mu) = NaN: every value is NaN, the first index wins, as in the reference
    idx, best = gp0.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
    ei = gp0.read(mu=False, var=False, acq=True)
    ei_o = o.expected_improvement(mu2, var, 0.01)
    assert np.all(np.isnan(ei)) and np.all(np.isnan(ei_o)) and idx == int(np.argmax(ei_o)) == 0 and np.isnan(best)
    # a NaN variance only: max(mu) stays finite, the first NaN acquisition value is picked
    var2 = var.copy()
    var2[3000] = np.nan
    gp0.set_posterior(mu, var2)
    idx, best = gp0.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
    assert idx == 3000 and np.isnan(best)


def test_failed_fit_leaves_wrapper_and_library_in_step():
    """gpbo_fit replaces n, d and drops candidates of another width before the factorisation can fail; the wrapper must
    hold the same shapes afterwards (it validates set_candidates / read against them)."""
    g = bo.GP(0)
    X, y = _data(50, 2, 5)
    g.fit(X, y, np.array([0.3, 0.4]), 0.05)
    g.set_candidates(o.splitmix64_uniform(3, 200).reshape(100, 2))
    g.posterior()
    X3 = o.splitmix64_uniform(4, 180).reshape(60, 3)
    X3[7, 1] = np.nan                                     # a NaN pivot is reported as not positive definite (LAPACK dpotrf rule)
    with pytest.raises(np.linalg.LinAlgError):
        g.fit(X3, np.ones(60), np.array([0.5, 0.5, 0.5]), 0.0)
    assert (g.n, g.d, g.m, g._xs_rows) == (60, 3, 0, 0)
    with pytest.raises(ValueError):
        g.set_candidates(o.splitmix64_uniform(3, 200).reshape(100, 2))      # two columns: the library now expects three
    with pytest.raises(ValueError):
        g.read()
    g.fit(X, y, np.array([0.3, 0.4]), 0.05)
    g.set_candidates(o.splitmix64_uniform(3, 200).reshape(100, 2))
    assert np.isfinite(g.posterior())
    g.close()
