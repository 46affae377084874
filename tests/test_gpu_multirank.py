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


def test_two_rank_sweep_equals_single_process(gp0):
    from bayesian_optimization_amd import synth
    m_per = 65536
    env = dict(os.environ, GPBO_DIST_BACKEND='gloo', GPBO_FORCE_DEVICE='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--candidates', str(m_per)]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['config']['candidates_per_gpu'] == m_per
    cfg = synth.CONFIGS['cfg2']
    X, y, theta = synth.training_set(cfg)
    gp0.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    gp0.set_candidates(synth.candidates(cfg, 0, 2 * m_per))
    mu_max = gp0.posterior()
    idx, best = gp0.acquire(0, cfg['kappa'])
    assert res['argmax']['index'] == idx and res['argmax']['ei'] == best and res['argmax']['mu_max'] == mu_max
    assert idx >= 0 and np.isfinite(best)
