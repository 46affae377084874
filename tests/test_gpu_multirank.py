"""GPU: two ranks (gloo rendezvous, both on cuda:0) run bench.py's sharded sweep; the global argmax, its EI and
max(Mu) must be bit-identical to one process sweeping the concatenated candidate set."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, env, tag):
    """Runs the launcher with stdout/stderr going to files (kept for diagnosis if the ranks stall), returns stdout."""
    logdir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(logdir, exist_ok=True)
    po, pe = os.path.join(logdir, tag + '.out'), os.path.join(logdir, tag + '.err')
    with open(po, 'w') as fo, open(pe, 'w') as fe:
        proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=fo, stderr=fe, timeout=float(os.environ.get('GPBO_TEST_TIMEOUT', '150')))
    assert proc.returncode == 0, open(pe).read()[-2000:]
    return open(po).read()


def test_two_rank_sweep_equals_single_process(gp0):
    from bayesian_optimization_amd import synth
    m_per = 65536
    env = dict(os.environ, GPBO_DIST_BACKEND='gloo', GPBO_FORCE_DEVICE='0', MASTER_ADDR='127.0.0.1', GPBO_STALL_DUMP='60')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    # no launcher on the command line: --gpus 2 makes bench.py start its two ranks itself
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--candidates', str(m_per), '--no-cfg3']
    out = _run(cmd, env, 'two_rank_sweep')
    line = [ln for ln in out.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['scaling'] == 'weak' and res['config']['candidates_per_gpu'] == m_per
    assert res['end_to_end']['broadcast_ms'] > 0 and res['end_to_end']['fit_ms'] > 0
    cfg = synth.CONFIGS['cfg2']
    X, y, theta = synth.training_set(cfg)
    gp0.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    gp0.set_candidates(synth.candidates(cfg, 0, 2 * m_per))
    mu_max = gp0.posterior()
    idx, best = gp0.acquire(0, cfg['kappa'])
    assert res['argmax']['index'] == idx and res['argmax']['ei'] == best and res['argmax']['mu_max'] == mu_max
    assert idx >= 0 and np.isfinite(best)


def test_two_rank_sharded_loop_picks_the_same_points(gp0):
    """BO loop with the lattice (and the resident V) split over two ranks: the sequence of chosen candidates equals the
    one-process loop's, index for index."""
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import synth, dist as gdist
    side, iters, d, n0 = 16, 12, 4, 16
    env = dict(os.environ, GPBO_DIST_BACKEND='gloo', GPBO_FORCE_DEVICE='0', MASTER_ADDR='127.0.0.1', GPBO_STALL_DUMP='60')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'cfg4', '--side', str(side),
           '--iters', str(iters), '--warmup', '0']
    out = _run(cmd, env, 'two_rank_loop')
    res = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['scaling'] == 'strong'
    g1 = (np.arange(side) + 0.5) / side
    grid = np.stack(np.meshgrid(g1, g1, g1, g1, indexing='ij'), -1).reshape(-1, d)
    f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
    X0 = synth.uniform(4, n0 * d).reshape(n0, d)
    gp0.reserve_appends(iters)
    gp0.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
    loop = gdist.ShardedLoop(gp0, grid, 0)
    picked = []
    for _ in range(iters):
        jj, _, x = loop.propose(bo.ACQ_EI, 0.01)
        assert np.array_equal(x, grid[jj])
        picked.append(jj)
        loop.observe(x, float(f(x)))
    assert res['config']['picked_head'] == picked
    assert res['config']['best_y'] == gp0.y_max


def test_sharded_lml_single_process_equals_lml_batch(gp0):
    from bayesian_optimization_amd import dist as gdist
    import oracle.gp_oracle as o
    X = o.splitmix64_uniform(77, 200 * 3).reshape(200, 3)
    y = np.sin(X.sum(1))
    th = np.array([[0.3, 0.4, 0.5], [0.2, 0.2, 0.2], [0.6, 0.5, 0.4]])
    sg = np.array([0.1, 0.05, 0.2])
    a = gp0.lml_batch(X, y, th, sg, grad=True)
    b = gdist.sharded_lml(gp0, X, y, th, sg, grad=True)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_rccl_code_path_with_a_single_rank():
    """One rank over the nccl (= RCCL) backend: the device-tensor exchanges of the sharded sweep (in-place broadcast of the
    model buffers through zero-copy views, all-reduce of max(Mu), all-gather of the argmax pairs) run through RCCL itself."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from bayesian_optimization_amd import GP, synth, dist as gdist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
gdist._is_dist = lambda: True                      # world_size 1 normally short-circuits the exchanges
cfg = dict(synth.CONFIGS['cfg2']); cfg['n'] = 512
X, y, theta = synth.training_set(cfg)
gp = GP(0); gp.fit(X, y, theta, cfg['sigmaY'])
Xs = synth.candidates(cfg, 0, 4096)
gp.set_candidates(Xs); mu0 = gp.posterior(); i0, b0 = gp.acquire(0, cfg['kappa'])
gdist.broadcast_model(gp, 0)                       # broadcasts W, alpha, Xts, params in place over RCCL
gidx, gbest, mu_max = gdist.ShardedSweep(gp, offset=0).step(0, cfg['kappa'])
assert (gidx, gbest, mu_max) == (i0, b0, mu0), ((gidx, gbest, mu_max), (i0, b0, mu0))
assert np.array_equal(gdist.allreduce_sum(np.arange(5.0), 0), np.arange(5.0))
dist.destroy_process_group()
print('RCCL_OK')
'''
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29549')
    logdir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(logdir, exist_ok=True)
    with open(os.path.join(logdir, 'rccl_single.out'), 'w') as fo, open(os.path.join(logdir, 'rccl_single.err'), 'w') as fe:
        proc = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, stdout=fo, stderr=fe, timeout=150)
    err = open(os.path.join(logdir, 'rccl_single.err')).read()
    assert proc.returncode == 0, err[-2000:]
    assert 'RCCL_OK' in open(os.path.join(logdir, 'rccl_single.out')).read()


def test_two_rank_strong_scaling_splits_one_candidate_set(gp0):
    """--scaling strong: ONE candidate set (odd size: the ranks' blocks differ by one) split over two ranks; argmax, EI and
    max(Mu) equal one process sweeping the whole set."""
    from bayesian_optimization_amd import synth
    m_total, n = 100001, 2048
    env = dict(os.environ, GPBO_DIST_BACKEND='gloo', GPBO_FORCE_DEVICE='0', MASTER_ADDR='127.0.0.1', GPBO_STALL_DUMP='60')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--config', 'cfg3',
           '--scaling', 'strong', '--candidates', str(m_total), '--n-train', str(n)]
    out = _run(cmd, env, 'two_rank_strong')
    res = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['scaling'] == 'strong'
    assert res['config']['candidates_total'] == m_total and res['config']['candidates_per_gpu'] in (m_total // 2, m_total // 2 + 1)
    cfg = dict(synth.CONFIGS['cfg3'], n=n)
    X, y, theta = synth.training_set(cfg)
    gp0.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    gp0.set_candidates(synth.candidates(cfg, 0, m_total))
    mu_max = gp0.posterior()
    idx, best = gp0.acquire(0, cfg['kappa'])
    assert res['argmax']['index'] == idx and res['argmax']['ei'] == best and res['argmax']['mu_max'] == mu_max
