#!/usr/bin/env python3
"""Wall time of repeated gpbo_fit calls against the device time the library reports: python3 scripts/fit_wall.py [n ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_optimization_amd import GP, synth
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]
gp = GP(0)
for n in sizes:
    X = synth.uniform(1, n * 4).reshape(n, 4)
    y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(101, n)
    th = np.full(4, 0.35)
    gp.fit(X, y, th, 0.1)
    ws = []
    for rep in range(8):
        t0 = time.perf_counter()
        gp.fit(X, y, th, 0.1)
        ws.append((time.perf_counter() - t0) * 1e3)
    t = gp.timings()
    print('n=%5d  wall %s ms (min %.3f)  device total %.3f  chol %.3f x%d  inverse %.3f  trsv %.3f  kbuild %.3f'
          % (n, ' '.join('%.2f' % w for w in ws[:4]), min(ws), t['fit_total_ms'], t['chol_ms'] / max(1, t['chol_count']), t['chol_count'],
             t['inverse_ms'], t['trsv_ms'], t.get('kbuild_ms', float('nan'))), flush=True)
