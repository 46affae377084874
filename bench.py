#!/usr/bin/env python3
"""Benchmark of the GP posterior + EI + argmax sweep (the hot path) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--candidates M] [--n-train N]
                    [--scaling weak|strong]

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) the script starts the N ranks itself, as fresh child
processes of `python -m torch.distributed.run`, before anything in this process has touched the GPU; under a launcher
it is one of the ranks.

A step = one full pass over this rank's resident candidate block: K(X*,X) tiles, posterior mean, variance
(W * Ks^T with fused column sum of squares on fp64 MFMA), max(Mu) [all-reduce], EI, argmax [all-gather].
The model is fitted once (rank 0) and broadcast before the timed region; candidates are resident in HBM.
value = candidates processed by all ranks / max-over-ranks time.  Rank 0 prints one JSON line.

The default line measures BASELINE.json's 4096-point configuration (cfg2) and carries more blocks: "cfg3", the
32768-point Matern-5/2 model of the 64 M-candidate configuration on a bounded slice of its candidates per GPU; with more
than one GPU "cfg3_strong", ONE fixed candidate set of that model split over the ranks, with the fit on rank 0 and the
hand-over of the model inside its end-to-end figure; "cfg4", 200 EI iterations with incremental updates on the 64^4
candidate lattice; "cfg5", 32 likelihood settings per GPU at n = 16384 (--no-cfg3 / --no-cfg45 leave them out).

--single-process drives the N GPUs from ONE process through the library's multi-context handle (gpbo_multi_*, the route
the notebooks take: no launcher, no torch.distributed) and prints the same line shape for the chosen configuration.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 flop/clk (v_mfma_f64_16x16x4: 2048 flop / 64 clk) x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def launch_ranks(n):
    """Start n ranks of this script (same arguments) under torch.distributed.run and return its exit code.  Called before
    torch or the HIP library is imported: this process never initialises the GPU, it only waits for the children."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(cfg, X, y, theta, Xs_sample):
    """The oracle restatement (NumPy + LAPACK threads) timed on the host cores, on a bounded sample."""
    from oracle import gp_oracle as o
    blas = 'unknown BLAS'
    try:
        from threadpoolctl import threadpool_info
        info = threadpool_info()
        cores = max([p.get('num_threads', 1) for p in info] + [1])
        libs = sorted({'%s %s' % (p.get('internal_api', '?'), p.get('version', '')) for p in info if p.get('user_api') == 'blas'})
        if libs:
            blas = ', '.join(libs)
    except Exception:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    st = o.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    mu, var = o.predict(st, Xs_sample, block=4096)
    ei = o.expected_improvement(mu, var, cfg['kappa'])
    int(np.argmax(ei))
    t_pred = time.perf_counter() - t0
    return dict(value=len(Xs_sample) / t_pred, unit='candidates/s', cores=int(cores), kind='port',
                sample='oracle/gp_oracle.py (NumPy/SciPy on %s, %d threads): predict+EI+argmax over the first %d candidates of '
                       'the same workload after an untimed-in-value fit of n=%d (fit %.2f s)' % (blas, cores, len(Xs_sample), cfg['n'], t_fit),
                fit_seconds=t_fit), st, (mu, var)


def _dist_setup(need_gpu=True):
    """(world, rank, device index, backend).  GPBO_DIST_BACKEND=gloo and GPBO_FORCE_DEVICE=0 allow a multi-rank
    rehearsal on a single-GPU box; the default is one rank per GPU over nccl (= RCCL)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev = int(os.environ['GPBO_FORCE_DEVICE']) if 'GPBO_FORCE_DEVICE' in os.environ else local_rank
    backend = os.environ.get('GPBO_DIST_BACKEND', 'nccl')
    if need_gpu:
        torch.cuda.set_device(dev)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
        else:
            dist.init_process_group(backend)
        if need_gpu:
            # open every pairwise connection the model hand-over and the per-sweep exchanges use before anything is timed
            # (the first point-to-point operation between two ranks sets the connection up: seconds, not microseconds)
            from bayesian_optimization_amd import dist as gdist
            w = torch.zeros(1 << 16, dtype=torch.float64, device='cuda')
            gdist.scatter_allgather(w, 0)
            dist.all_reduce(w[:1])
            dist.broadcast(w[:8], 0)
            torch.cuda.synchronize()
    return world, rank, dev, backend


def _max_over_ranks(dt, world, backend):
    import torch
    import torch.distributed as dist
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return dt


def _ranks_seen(world):
    import torch.distributed as dist
    return dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1


def _profiled_traffic(config, n, cand_per_launch):
    """Bytes per launch of the dominant kernel that crossed the L2 / fabric boundary, from a committed PMC pass of the same
    workload (scripts/profile_bench.sh: separate --pmc runs, FETCH_SIZE doubled as the gfx950 counter needs).  Counters cannot
    be read from inside this process: roofline.traffic carries the profiled figure, roofline.traffic_source says where from."""
    for name in ('r3_%s_traffic.json' % config, 'r2_%s_traffic.json' % config, 'r1_%s_traffic.json' % config):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            tr = json.load(open(path))
            if tr['n_train'] == n and abs(cand_per_launch - tr['candidates_per_launch']) < 1:
                return {'bytes_per_launch': tr['fetch_bytes_per_launch'] + tr['write_bytes_per_launch'], 'source': 'profiles/' + name,
                        'algorithmic': tr.get('algorithmic_bytes_per_launch'), 'l2_hit_rate': tr.get('l2_hit_rate')}
        except Exception:
            continue
    return None


def sweep_block(a, config, gp, world, rank, backend, steps, warmup, with_cpu):
    """Fit (rank 0) -> broadcast -> resident candidates -> `warmup` + `steps` sweeps.  Returns the result dict on rank 0."""
    import torch
    import torch.distributed as dist
    from bayesian_optimization_amd import synth
    from bayesian_optimization_amd import dist as gdist

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = dict(synth.CONFIGS[config])
    if a.n:
        cfg['n'] = a.n
    strong = a.scaling == 'strong'
    if strong:
        m_total = a.m if a.m else cfg['m']
        lo, hi = gdist.shard_range(m_total, rank, world)
    else:
        m_per_gpu = a.m if a.m else cfg['m']
        if config == 'cfg3' and not a.m:
            m_per_gpu = 1 << 17 if a.config != 'cfg3' else 1 << 18      # a bounded slice per GPU (stated in config.workload)
        lo, hi = rank * m_per_gpu, (rank + 1) * m_per_gpu
        m_total = world * m_per_gpu
    m_local = hi - lo

    X, y, theta = synth.training_set(cfg)
    fit_tm, cond_est, fit_wall, fit_first = None, None, 0.0, 0.0
    barrier()
    t0 = time.perf_counter()
    if rank == 0:
        gp.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
        fit_wall = fit_first = time.perf_counter() - t0
        if cfg['n'] <= 8192:
            # the first call of a process also allocates the model's buffers and loads the kernels (milliseconds, which
            # show at this size only): the per-phase times are those of a second call, the first call's wall time is
            # kept beside them and stays the one the end-to-end figure uses
            t1 = time.perf_counter()
            gp.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
            fit_wall = time.perf_counter() - t1
        fit_tm = gp.timings()
        cond_est = gp.cond_estimate()
    barrier()
    t0 = time.perf_counter()
    gdist.broadcast_model(gp, 0)
    barrier()
    bcast_s = _max_over_ranks(time.perf_counter() - t0, world, backend) if world > 1 else 0.0
    t0 = time.perf_counter()
    gp.set_candidates_uniform(cfg['seed_c'], lo, m_local)          # generated in HBM: same numbers as synth.candidates(cfg, lo, hi)
    cand_s = time.perf_counter() - t0
    sweep = gdist.ShardedSweep(gp, offset=lo)

    for _ in range(warmup):
        sweep.step(0, cfg['kappa'])
    barrier()
    t0 = time.perf_counter()
    trmm_ms = trmm_launches = ks_ms = post_ms = acq_ms = 0.0
    for _ in range(steps):
        gidx, gbest, mu_max = sweep.step(0, cfg['kappa'])
        tm = gp.timings()                      # HIP-event times on the library's own stream, per step
        trmm_ms += tm['trmm_ms']
        trmm_launches += tm['trmm_launches']
        ks_ms += tm['ksbuild_ms']
        post_ms += tm['posterior_total_ms']
        acq_ms += tm['acquire_ms']
    barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, world, backend)
    if rank != 0:
        return None

    n = cfg['n']
    total = m_total * steps
    cand_per_launch = m_local * steps / max(trmm_launches, 1)
    flops_per_launch = float(n) * n * cand_per_launch             # n^2 per candidate x candidates per launch
    avg_launch_s = trmm_ms * 1e-3 / max(trmm_launches, 1)
    achieved = flops_per_launch / avg_launch_s / 1e12
    prof = _profiled_traffic(config, n, cand_per_launch)
    out = {
        'metric': 'gp_posterior_ei_candidates_per_sec', 'value': total / dt, 'unit': 'candidates/s',
        'n_gpus': world, 'ranks_seen': _ranks_seen(world), 'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: n=%d train, d=%d, %s, %s, sigmaY=%g, theta=%g, EI kappa=%g'
                               % (config, n, cfg['d'], 'RBF' if cfg['kernel'] == 0 else 'Matern-5/2',
                                  ('%d candidates split over %d GPUs' % (m_total, world)) if strong else
                                  ('%d candidates per GPU (x%d GPUs)' % (m_local, world)),
                                  cfg['sigmaY'], cfg['theta'], cfg['kappa']),
                   'candidates_per_gpu': m_local, 'candidates_total': m_total, 'n_train': n, 'd': cfg['d'],
                   'parallelism': 'candidate blocks per rank; all-reduce(max mu) + all-gather(argmax)'},
        'roofline': {'bound': 'mfma', 'kernel': 'k_trmm_sumsq', 'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': achieved / FP64_MFMA_PEAK_TFLOPS,
                     'traffic': prof['bytes_per_launch'] if prof else None,
                     'traffic_source': ('%s: rocprofv3 PMC passes of this workload with this kernel (L2 fills + writes per launch, not '
                                        'measured in this run; algorithmic %.4g bytes, L2 hit rate %s)'
                                        % (prof['source'], prof['algorithmic'] or 0.0, prof['l2_hit_rate'])) if prof else 'not measured in this run',
                     'flops_per_launch': flops_per_launch, 'avg_launch_ms': avg_launch_s * 1e3, 'launches': trmm_launches},
        'phases_ms_per_step': {'posterior_total': post_ms / steps, 'ks_build': ks_ms / steps,
                               'trmm_sumsq': trmm_ms / steps, 'acquire': acq_ms / steps},
        'argmax': {'index': int(gidx), 'ei': gbest, 'mu_max': mu_max},
    }
    # every phase against its own bound, and the time-weighted combination over the sweep
    ks_s, other_s = ks_ms * 1e-3, max(post_ms + acq_ms - ks_ms - trmm_ms, 0.0) * 1e-3
    cand = float(m_local * steps)
    ks_gbs = 8.0 * n * cand / max(ks_s, 1e-12) / 1e9                   # the Ks panel: 8 n bytes written per candidate
    other_gbs = (8.0 * cfg['d'] + 48.0) * cand / max(other_s, 1e-12) / 1e9   # read x*, write mu/var/ei, EI pass re-reads mu/var
    ph = [('trmm_sumsq', 'mfma', trmm_ms * 1e-3, achieved / FP64_MFMA_PEAK_TFLOPS),
          ('ks_build', 'hbm', ks_s, ks_gbs / HBM_PEAK_GBS), ('finalize+acquire+reductions', 'hbm', other_s, other_gbs / HBM_PEAK_GBS)]
    tsum = sum(t for _, _, t, _ in ph)
    out['phase_rooflines'] = {'phases': [{'phase': k, 'bound': b, 'ms_per_step': t / steps * 1e3, 'frac': f} for k, b, t, f in ph],
                              'time_weighted_frac': sum(t * f for _, _, t, f in ph) / max(tsum, 1e-12)}
    nchol = max(fit_tm['chol_count'], 1)
    out['fit'] = {'total_ms': fit_tm['fit_total_ms'], 'wall_ms': fit_wall * 1e3, 'first_call_wall_ms': fit_first * 1e3,
                  'cholesky_ms_each': fit_tm['chol_ms'] / nchol,
                  'cholesky_gflops': (n ** 3 / 3.0) / (fit_tm['chol_ms'] / nchol * 1e-3) / 1e9,
                  'inverse_ms': fit_tm['inverse_ms'], 'trsv_ms': fit_tm['trsv_ms'], 'kbuild_ms': fit_tm['kbuild_ms'],
                  'cond_estimate': cond_est}
    sweep_s = dt / steps
    out['end_to_end'] = {'fit_ms': fit_first * 1e3, 'broadcast_ms': bcast_s * 1e3, 'candidates_ms': cand_s * 1e3,
                         'sweep_ms': sweep_s * 1e3,
                         'broadcast_bytes': 8.0 * ((-(-n // 128) * 128) ** 2) if world > 1 else 0.0,
                         'candidates_per_sec_incl_fit_and_broadcast': m_total / (fit_first + bcast_s + sweep_s)}
    if with_cpu:
        ns = 16384 if n <= 8192 else 2048
        Xs = synth.candidates(cfg, lo, lo + ns)
        base, st, (mu_o, var_o) = cpu_baseline(cfg, X, y, theta, Xs)
        out['cpu_baseline'] = base
        mu, var = gp.read(offset=0, count=ns)
        rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
        out['parity_vs_oracle'] = {'sample': ns, 'mean_relmax': rel(mu, mu_o), 'var_relmax': rel(var, var_o)}
    return out


def cfg4_block(a, gp, world, rank, backend, iters, warmup, lattice=False):
    """cfg4: BayesOpt_v4-style loop, theta frozen: EI iterations with incremental append on a 64^4 candidate lattice
    (a step = one iteration: acquisition + argmax + border-append + the update of every candidate's mean and variance).
    lattice=False: the candidates are handed over as a plain array of points and the update streams the resident V (what
    any candidate set gets).  lattice=True: they are handed over as a lattice (per-axis coordinates) and the update is one
    product of per-axis kernel factors on the MFMA pipe followed by the same read-modify-write of mean / variance.
    Returns the JSON dictionary on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    import bayesian_optimization_amd as bo
    from bayesian_optimization_amd import synth
    from bayesian_optimization_amd import dist as gdist

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    d, side, n0 = 4, a.side, 16
    g1 = (np.arange(side) + 0.5) / side
    m_total = side ** d
    sharded = world > 1 and not a.replicas       # candidate rows (and the resident V) split over the ranks: one loop, strong scaling
    lo, hi = gdist.shard_range(m_total, rank, world) if sharded else (0, m_total)
    # lexicographic order, as the notebooks' nested loops; built per block from the flat index
    ii = np.arange(lo, hi) if not lattice else np.arange(0)
    grid = np.stack([g1[(ii // side ** (d - 1 - k)) % side] for k in range(d)], -1)
    f = lambda x: np.exp(-8 * np.sum((x - 0.37) ** 2, -1)) + 0.5 * np.exp(-20 * np.sum((x - 0.8) ** 2, -1))
    X0 = synth.uniform(4 if sharded else 4 + rank, n0 * d).reshape(n0, d)     # replicas: every rank runs its own independent loop
    gp.reserve_appends(iters + warmup)
    gp.fit(X0, f(X0), np.full(d, 0.25), 1e-2, variant=bo.VARIANT_SINGLE)
    loop = gdist.ShardedLoop.on_lattice(gp, [g1] * d, exchange=sharded) if lattice else gdist.ShardedLoop(gp, grid, lo, exchange=sharded)
    m = gp.m
    sweep_ms, bytes_alg = 0.0, 0.0
    picked = []
    # an iteration = append the point picked last + sweep + pick the next one (ShardedLoop.step)
    jj, _, x = loop.propose(bo.ACQ_EI, 0.01)
    for it in range(warmup):
        picked.append(jj)
        jj, _, x = loop.step(x, float(f(x)), bo.ACQ_EI, 0.01)
    barrier()
    t0 = time.perf_counter()
    timed = 0
    for it in range(iters):
        picked.append(jj)
        n_before = gp.n
        sample = it % 8 == 0           # the sweep kernel is timed (HIP events) on every eighth iteration: the events cost ~1 % of a step
        gp.set_timing(sample)
        jj, ei, x = loop.step(x, float(f(x)), bo.ACQ_EI, 0.01)
        if sample:
            sweep_ms += gp.timings()['append_sweep_ms']
            if lattice:       # mean r/w, sum v^2 r/w (the variance is formed from it on use); the two factor matrices ((P + Q)(n + 1) doubles) once
                bytes_alg += m * 32.0 + 8.0 * (n_before + 1) * (side ** ((d + 1) // 2) + side ** (d // 2))
            else:             # V read + new V row + (mu, var r/w; sum v^2 r/w) + candidates
                bytes_alg += m * (8.0 * n_before + 8 + 32 + 8 * d)
            timed += 1
    gp.set_timing(True)
    barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, world, backend)
    if rank != 0:
        return None
    achieved = bytes_alg / (max(sweep_ms, 1e-9) * 1e-3) / 1e9
    loops = 1 if sharded else world
    return {
        'metric': 'bo_loop_iterations_per_sec', 'value': loops * iters / dt, 'unit': 'iterations/s', 'n_gpus': world,
        'ranks_seen': _ranks_seen(world),
        'steps': iters, 'warmup': warmup, 'ms_per_step': dt / iters * 1e3, 'higher_is_better': True,
        'scaling': 'strong' if sharded else 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'cfg4: %d EI iterations, incremental append (theta frozen, single-factor posterior), '
                               '%d^4 = %d candidate lattice, d=4, n0=%d, sigmaY=1e-2; %s'
                               % (iters, side, m_total, n0, ('candidate rows and resident V split over the GPUs, one loop'
                                  if sharded else 'one independent loop per GPU') +
                                  ('; candidates given as a lattice: per-append update = one (%d x %d x n) product of per-axis kernel '
                                   'factors, no resident V' % (side ** ((d + 1) // 2), side ** (d // 2)) if lattice else
                                   '; candidates given as points: per-append update streams the resident V')),
                   'n_final': gp.n, 'best_y': float(gp.y_max), 'picked_head': [int(j) for j in picked[:16]]},
        'roofline': {'bound': 'hbm', 'kernel': 'k_lattice_gemm_update_acq (update and acquisition of all candidates in one launch; + factor build)' if lattice else 'k_append_sweep', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'avg_launch_ms': sweep_ms / max(timed, 1),
                     'bytes_per_launch_avg': bytes_alg / max(timed, 1), 'launches_timed': timed}}


def cfg5_block(a, gp, world, rank, backend, per, steps, warmup):
    """cfg5: (length-scale, noise) settings of -logL + analytic gradient at n = 16384, `per` of the 256 per GPU (a step =
    the whole sweep of this rank's share; ranks take disjoint settings, no exchange).  Rank 0 returns the dictionary."""
    import torch
    import torch.distributed as dist
    from bayesian_optimization_amd import synth

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n, d = (a.n or 16384), 8
    Btot = 256
    X = synth.uniform(5, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1)) + 0.1 * synth.normal(105, n)
    mult = np.geomspace(0.25, 2.0, 16)
    sigs = np.geomspace(1e-2, 1.0, 16)
    allset = [(mm, ss) for mm in mult for ss in sigs]
    mine = [allset[(rank * per + i) % Btot] for i in range(per)]
    thetas = np.array([np.full(d, 0.5 * mm) for mm, _ in mine])
    sig = np.array([ss for _, ss in mine])
    for _ in range(warmup):
        gp.lml_batch(X, y, thetas[:2], sig[:2], grad=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        nll, grad, stat = gp.lml_batch(X, y, thetas, sig, grad=True)
    barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, world, backend)
    if rank != 0:
        return None
    flops = float(n) ** 3 * per * steps * world       # n^3/3 Cholesky + n^3/3 inverse + n^3/3 W^T W per setting
    achieved = flops / dt / 1e12 / world
    return {
        'metric': 'lml_grad_settings_per_sec', 'value': world * per * steps / dt, 'unit': 'settings/s', 'n_gpus': world,
        'ranks_seen': _ranks_seen(world),
        'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'cfg5: -logL + analytic gradient, n=%d, d=%d, RBF, %d settings per GPU (16 length-scale multipliers x 16 sigmaY)' % (n, d, per),
                   'failed_settings': int(np.sum(stat != 0))},
        'roofline': {'bound': 'mfma', 'kernel': 'whole sweep (Cholesky + inverse + W^T W tiles)', 'achieved': achieved,
                     'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP64_MFMA_PEAK_TFLOPS, 'traffic': None}}


def single_process(a):
    """--single-process: one process, one multi-context handle over a.gpus GPUs (gpbo_multi_*).  A step = the whole
    candidate set of the configuration, split into one block per GPU inside the library: posterior on every block side by
    side, host-side max(Mu), acquisition on every block, host-side argmax."""
    from bayesian_optimization_amd import MultiGP, synth
    cfg = dict(synth.CONFIGS[a.config])
    if a.n:
        cfg['n'] = a.n
    m_total = a.m if a.m else cfg['m']
    if a.config == 'cfg3' and not a.m:
        m_total = (1 << 18) * a.gpus          # a bounded slice per GPU (stated in config.workload)
    devs = [int(v) for v in os.environ['GPBO_DEVICES'].split(',')] if os.environ.get('GPBO_DEVICES') else list(range(a.gpus))
    gp = MultiGP(devs)
    if a.chunk:
        gp.set_chunk(a.chunk)
    X, y, theta = synth.training_set(cfg)
    t0 = time.perf_counter()
    gp.fit(X, y, theta, cfg['sigmaY'], kernel=cfg['kernel'])
    fit_s = time.perf_counter() - t0
    dist_ms, dist_bytes = gp.last_distribute()
    gp.set_candidates_uniform(cfg['seed_c'], 0, m_total)
    for _ in range(a.warmup):
        gp.posterior()
        gp.acquire(0, cfg['kappa'])
    t0 = time.perf_counter()
    trmm_ms = [0.0] * len(devs)
    launches = [0.0] * len(devs)
    for _ in range(a.steps):
        mu_max = gp.posterior()
        gidx, gbest = gp.acquire(0, cfg['kappa'])
        for i, tm in enumerate(gp.timings()):
            trmm_ms[i] += tm['trmm_ms']
            launches[i] += tm['trmm_launches']
    dt = time.perf_counter() - t0
    n = cfg['n']
    blocks = gp.blocks()
    per_ctx = [float(n) * n * (hi - lo) * a.steps / max(ms * 1e-3, 1e-12) / 1e12 for (lo, hi), ms in zip(blocks, trmm_ms)]
    slow = int(np.argmin(per_ctx))
    cand_per_launch = (blocks[slow][1] - blocks[slow][0]) * a.steps / max(launches[slow], 1)
    out = {
        'metric': 'gp_posterior_ei_candidates_per_sec', 'value': m_total * a.steps / dt, 'unit': 'candidates/s',
        'n_gpus': len(devs), 'processes': 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: n=%d train, d=%d, %s, %d candidates split over %d contexts (devices %s) in one process, '
                               'sigmaY=%g, theta=%g, EI kappa=%g'
                               % (a.config, n, cfg['d'], 'RBF' if cfg['kernel'] == 0 else 'Matern-5/2', m_total, len(devs),
                                  ','.join(map(str, devs)), cfg['sigmaY'], cfg['theta'], cfg['kappa']),
                   'candidates_total': m_total, 'n_train': n, 'd': cfg['d'],
                   'parallelism': 'one process, one context + host thread per GPU; host-side max(mu) and argmax'},
        'roofline': {'bound': 'mfma', 'kernel': 'k_trmm_sumsq', 'achieved': per_ctx[slow], 'peak': FP64_MFMA_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': per_ctx[slow] / FP64_MFMA_PEAK_TFLOPS, 'traffic': None,
                     'context': slow, 'per_context_tflops': per_ctx, 'flops_per_launch': float(n) * n * cand_per_launch,
                     'avg_launch_ms': trmm_ms[slow] / max(launches[slow], 1), 'launches': launches[slow]},
        'end_to_end': {'fit_and_distribute_ms': fit_s * 1e3, 'distribute_ms': dist_ms, 'distribute_bytes': dist_bytes,
                       'sweep_ms': dt / a.steps * 1e3,
                       'candidates_per_sec_incl_fit_and_distribute': m_total / (fit_s + dt / a.steps)},
        'argmax': {'index': int(gidx), 'ei': gbest, 'mu_max': mu_max},
    }
    print(json.dumps(out), flush=True)
    gp.close()


def secondary(a):
    """The two other GPU configurations of BASELINE.json on their own, same launch contract and JSON shape."""
    import torch.distributed as dist
    from bayesian_optimization_amd import GP
    world, rank, local_rank, backend = _dist_setup()
    gp = GP(local_rank)
    if a.config == 'cfg4':
        out = cfg4_block(a, gp, world, rank, backend, a.iters, a.warmup, lattice=a.lattice)
    else:
        out = cfg5_block(a, gp, world, rank, backend, a.settings if a.settings else max(1, 256 // world), a.steps, a.warmup)
    if rank == 0:
        print(json.dumps(out), flush=True)
    gp.close()
    if world > 1:
        dist.destroy_process_group()


def rehearse(a):
    """Launch plumbing only, no GPU: the ranks rendezvous (gloo), split a candidate range and run the sweep's two exchange
    steps on stand-in local results (tests/_standin_gp.py).  Lets a CPU-only machine check that --gpus N really runs N ranks."""
    import torch.distributed as dist
    from bayesian_optimization_amd import dist as gdist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _standin_gp import StandinGP
    os.environ.setdefault('GPBO_DIST_BACKEND', 'gloo')
    world, rank, _, backend = _dist_setup(need_gpu=False)
    m_total = a.m if a.m else 1000
    lo, hi = gdist.shard_range(m_total, rank, world)
    sweep = gdist.ShardedSweep(StandinGP(lo, hi), offset=lo)
    gidx, gbest, mu_max = sweep.step(0, 0.01)
    if rank == 0:
        print(json.dumps({'rehearsal': True, 'n_gpus': world, 'ranks_seen': _ranks_seen(world), 'backend': backend,
                          'argmax': {'index': int(gidx), 'ei': gbest, 'mu_max': mu_max}, 'value': None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--candidates', dest='m', type=int, default=0,
                    help='candidates per GPU (weak scaling) or in total (strong scaling); default: the config\'s m')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: every rank sweeps the same number of candidates; strong: the set is split over the ranks')
    ap.add_argument('--n-train', dest='n', type=int, default=0, help='training points (default: the config\'s n)')
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cfg3', action='store_true', help='leave the 32768-point block out of the default line')
    ap.add_argument('--no-cfg45', action='store_true', help='leave the lattice-loop and likelihood-sweep blocks out of the default line')
    ap.add_argument('--iters', type=int, default=200, help='cfg4: EI iterations')
    ap.add_argument('--side', type=int, default=64, help='cfg4: points per axis of the candidate lattice')
    ap.add_argument('--lattice', action='store_true', help='cfg4: hand the candidates over as a lattice (per-axis coordinates)')
    ap.add_argument('--replicas', action='store_true', help='cfg4 with several GPUs: independent loops instead of one sharded loop')
    ap.add_argument('--settings', type=int, default=0, help='cfg5: settings per GPU (default 256 / gpus)')
    ap.add_argument('--rehearse', action='store_true', help='launch/exchange plumbing only, on the CPU (no GPU work, no measurement)')
    ap.add_argument('--single-process', action='store_true',
                    help='drive the --gpus GPUs from this one process through the multi-context handle (cfg2 / cfg3 sweep)')
    a = ap.parse_args()
    if a.single_process and 'WORLD_SIZE' not in os.environ:
        if a.config not in ('cfg2', 'cfg3'):
            sys.exit('bench.py: --single-process measures the candidate sweep (cfg2 / cfg3)')
        return single_process(a)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != a.gpus:
        print('bench.py: --gpus %d but the launcher started %s ranks; reporting the launcher\'s count'
              % (a.gpus, os.environ['WORLD_SIZE']), file=sys.stderr)
    if os.environ.get('GPBO_STALL_DUMP'):      # diagnosis: dump every thread's Python stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['GPBO_STALL_DUMP']), exit=False)
    if a.rehearse:
        return rehearse(a)
    if a.config in ('cfg4', 'cfg5'):
        return secondary(a)

    import torch.distributed as dist
    from bayesian_optimization_amd import GP

    world, rank, local_rank, backend = _dist_setup()
    gp = GP(local_rank)
    if a.chunk:
        gp.set_chunk(a.chunk)
    out = sweep_block(a, a.config, gp, world, rank, backend, a.steps, a.warmup, with_cpu=(world == 1 and not a.no_cpu_baseline))
    if a.config == 'cfg2' and not a.no_cfg3 and not a.n and not a.m:
        # the 64 M-candidate configuration's model (n = 32768, d = 8, Matern-5/2) on a 131072-candidate slice per GPU; the
        # CPU oracle is not run here (its fit alone takes over a minute): tests/ compare this model with the oracle
        gp.set_chunk(0)
        blk = sweep_block(a, 'cfg3', gp, world, rank, backend, steps=2, warmup=1, with_cpu=False)
        if rank == 0:
            out['cfg3'] = {k: blk[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'config', 'roofline',
                                               'phases_ms_per_step', 'phase_rooflines', 'fit', 'end_to_end', 'argmax')}
        if world > 1:
            # the north-star split: ONE candidate set of that model (a bounded 524288-candidate slice of the 64 M) divided
            # over the ranks; rank 0 fits, the model goes out over every link, and the end-to-end figure carries both
            a2 = argparse.Namespace(**vars(a))
            a2.scaling, a2.m = 'strong', 1 << 19
            blk = sweep_block(a2, 'cfg3', gp, world, rank, backend, steps=1, warmup=0, with_cpu=False)
            if rank == 0:
                out['cfg3_strong'] = {k: blk[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'config',
                                                          'roofline', 'phases_ms_per_step', 'fit', 'end_to_end', 'argmax')}
    gp.close()
    if a.config == 'cfg2' and not a.no_cfg45 and not a.n and not a.m:
        # the two remaining GPU configurations, each in a context of its own: the 64^4-lattice loop (200 iterations) and 32
        # likelihood settings per GPU at n = 16384
        keys = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'config', 'roofline')
        gp4 = GP(local_rank)
        blk = cfg4_block(a, gp4, world, rank, backend, 200, 1)
        gp4.close()
        if rank == 0:
            out['cfg4'] = {k: blk[k] for k in keys}
        gp4 = GP(local_rank)
        blk2 = cfg4_block(a, gp4, world, rank, backend, 200, 1, lattice=True)
        gp4.close()
        if rank == 0:
            out['cfg4_lattice'] = {k: blk2[k] for k in keys}
            out['cfg4_lattice']['same_picks_as_cfg4'] = blk2['config']['picked_head'] == blk['config']['picked_head']
        gp5 = GP(local_rank)
        blk = cfg5_block(a, gp5, world, rank, backend, 32, 1, 1)
        gp5.close()
        if rank == 0:
            out['cfg5'] = {k: blk[k] for k in keys}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
