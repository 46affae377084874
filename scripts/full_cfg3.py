#!/usr/bin/env python3
"""The 64 M-candidate configuration at full size on ONE GPU, once:  n = 32768, d = 8, Matern-5/2, 67,108,864 candidates.

    GPBO_PROGRESS=1 python3 scripts/full_cfg3.py [--candidates M] [--out gpurun_out/r2_cfg3_full]

No torch (the run has 20 minutes, the import would take two of them): NumPy + the ctypes binding only.  Candidates are
generated in HBM from the same counter stream bayesian_optimization_amd/synth.py defines.  Writes <out>.json (one line,
bench.py's shape) and <out>_sample.npz: mean / variance / EI on a fixed strided subsample of 4096 candidates and on the 16
candidates with the largest EI, for the comparison with the CPU oracle (oracle/make_oracle_fixtures.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 78.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--candidates', type=int, default=1 << 26)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r2_cfg3_full'))
    a = ap.parse_args()
    from bayesian_optimization_amd import GP, synth
    cfg = synth.CONFIGS['cfg3']
    m = a.candidates
    X, y, theta = synth.training_set(cfg)
    gp = GP(0)
    t0 = time.perf_counter()
    gp.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    fit_s = time.perf_counter() - t0
    ft = gp.timings()
    cond = gp.cond_estimate()
    print('[full] fit %.3f s, theta1 %.12g, cond estimate %.4g' % (fit_s, gp.theta1, cond), file=sys.stderr, flush=True)
    gp.set_candidates_uniform(cfg['seed_c'], 0, m)
    t0 = time.perf_counter()
    mu_max = gp.posterior()
    post_s = time.perf_counter() - t0
    tm = gp.timings()
    t1 = time.perf_counter()
    idx, best = gp.acquire(0, cfg['kappa'])
    acq_s = time.perf_counter() - t1
    sweep_s = post_s + acq_s
    print('[full] sweep %.1f s: argmax %d, EI %.15g, max(mu) %.15g' % (sweep_s, idx, best, mu_max), file=sys.stderr, flush=True)
    # read back in blocks; keep the strided subsample and the running top-16 by EI
    stride = max(1, m // 4096)
    sub = np.arange(0, m, stride)[:4096]
    mu_s, var_s, ei_s = np.empty(len(sub)), np.empty(len(sub)), np.empty(len(sub))
    top_i, top_v = np.empty(0, dtype=np.int64), np.empty(0)
    blk = 1 << 22
    var_min, var_max, nz = np.inf, -np.inf, 0
    for lo in range(0, m, blk):
        cnt = min(blk, m - lo)
        mu, var, ei = gp.read(acq=True, offset=lo, count=cnt)
        sel = sub[(sub >= lo) & (sub < lo + cnt)]
        k = np.searchsorted(sub, sel)
        mu_s[k], var_s[k], ei_s[k] = mu[sel - lo], var[sel - lo], ei[sel - lo]
        part = np.argpartition(ei, -16)[-16:]
        top_i = np.concatenate([top_i, part + lo])
        top_v = np.concatenate([top_v, ei[part]])
        order = np.lexsort((top_i, -top_v))[:16]          # value descending, index ascending on ties
        top_i, top_v = top_i[order], top_v[order]
        var_min, var_max = min(var_min, float(var.min())), max(var_max, float(var.max()))
        nz += int(np.count_nonzero(~np.isfinite(ei)))
    assert top_i[0] == idx and top_v[0] == best, (top_i[0], idx)      # the device argmax is the first maximum of the EI array
    mu_t, var_t, ei_t = np.empty(16), np.empty(16), np.empty(16)
    for q, j in enumerate(top_i):
        a1, b1, c1 = gp.read(acq=True, offset=int(j), count=1)
        mu_t[q], var_t[q], ei_t[q] = a1[0], b1[0], c1[0]
    np.savez_compressed(a.out + '_sample.npz', sub=sub, mu=mu_s, var=var_s, ei=ei_s, top_index=top_i, top_mu=mu_t, top_var=var_t,
                        top_ei=ei_t, mu_max=np.array(mu_max), theta1=np.array(gp.theta1), nll=np.array(gp.nll), m=np.array(m))
    n = cfg['n']
    launches = tm['trmm_launches']
    achieved = float(n) * n * m / (tm['trmm_ms'] * 1e-3) / 1e12
    nchol = max(ft['chol_count'], 1)
    out = {'metric': 'gp_posterior_ei_candidates_per_sec', 'value': m / sweep_s, 'unit': 'candidates/s', 'n_gpus': 1, 'steps': 1, 'warmup': 0,
           'ms_per_step': sweep_s * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'cfg3 at full size: n=%d train, d=%d, Matern-5/2, %d candidates on one GPU (generated in HBM), sigmaY=%g, theta=%g, EI kappa=%g'
                                  % (n, cfg['d'], m, cfg['sigmaY'], cfg['theta'], cfg['kappa']), 'candidates_total': m, 'n_train': n},
           'roofline': {'bound': 'mfma', 'kernel': 'k_trmm_sumsq', 'achieved': achieved, 'peak': PEAK, 'unit': 'TFLOP/s', 'frac': achieved / PEAK,
                        'traffic': None, 'flops_per_launch': float(n) * n * m / launches, 'avg_launch_ms': tm['trmm_ms'] / launches, 'launches': launches},
           'phases_ms': {'posterior_wall': post_s * 1e3, 'posterior_stream': tm['posterior_total_ms'], 'ks_build': tm['ksbuild_ms'],
                         'trmm_sumsq': tm['trmm_ms'], 'finalize': tm['finalize_ms'], 'acquire_wall': acq_s * 1e3},
           'whole_sweep_frac': float(n) * n * m / sweep_s / 1e12 / PEAK,
           'fit': {'wall_ms': fit_s * 1e3, 'total_ms': ft['fit_total_ms'], 'cholesky_ms_each': ft['chol_ms'] / nchol,
                   'cholesky_gflops': (n ** 3 / 3.0) / (ft['chol_ms'] / nchol * 1e-3) / 1e9, 'inverse_ms': ft['inverse_ms'], 'cond_estimate': cond},
           'argmax': {'index': int(idx), 'ei': best, 'mu_max': mu_max},
           'checks': {'var_min': var_min, 'var_max': var_max, 'theta1': gp.theta1, 'nonfinite_ei': nz}}
    with open(a.out + '.json', 'w') as f:
        f.write(json.dumps(out) + '\n')
    print(json.dumps(out), flush=True)
    gp.close()


if __name__ == '__main__':
    main()
