"""CPU: `python bench.py --gpus 2` with no launcher on the command line must start two ranks by itself (gloo rehearsal,
stand-in local results) and report what the ranks agreed on."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from _standin_gp import acq_of, mu_of


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, proc.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_launches_the_ranks_itself():
    m = 1001
    res = _run(['--gpus', '2', '--rehearse', '--candidates', str(m)])
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['backend'] == 'gloo'
    j = np.arange(m)
    mu_max = float(mu_of(j).max())
    a = acq_of(j, mu_max)
    assert res['argmax'] == {'index': int(np.argmax(a)), 'ei': float(a.max()), 'mu_max': mu_max}
    assert res['argmax']['index'] == 100


def test_three_ranks_and_single_rank_agree():
    m = 777
    r1 = _run(['--gpus', '1', '--rehearse', '--candidates', str(m)])
    r3 = _run(['--gpus', '3', '--rehearse', '--candidates', str(m)])
    assert r1['n_gpus'] == 1 and r3['n_gpus'] == 3 and r3['ranks_seen'] == 3
    assert r1['argmax'] == r3['argmax']
