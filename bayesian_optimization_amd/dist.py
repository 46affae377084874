"""Multi-GPU sharding of the candidate sweep: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in CPU tests).

Candidates are split into contiguous blocks (global index = block offset + local index, so the
first-occurrence argmax survives sharding).  The data path has exactly two exchange steps:
  1. all-reduce(MAX) of one double: the EI incumbent max(Mu) (BayesOpt/BayesOpt_v4.ipynb:193);
  2. all-gather of (acquisition max, global index) per rank, reduced to max value / lowest index.
The fitted model (inverse factor W, alpha, scaled inputs, scalars) is computed on rank `src`
only and handed over once per fit: the small buffers by broadcast, W by scatter + all-gather over every link.
(One process driving several GPUs takes the route of gp.MultiGP / gpbo_multi_* instead; this module is the
one-process-per-GPU form that torchrun launches.)
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(m_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank; the first m_total % world ranks get one extra item."""
    q, r = divmod(int(m_total), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _exchange_device(device):
    """Device on which the small exchange tensors live: the GPU under nccl, the host under gloo."""
    if _is_dist() and dist.get_backend() == 'nccl':
        return torch.device('cuda', device if device is not None else torch.cuda.current_device())
    return torch.device('cpu')


def allreduce_max(x: float, device=None) -> float:
    """Maximum over ranks with numpy.max's NaN rule (a NaN anywhere gives NaN): what a collective's MAX does with a NaN
    is not specified, so the NaN travels as a flag beside the value."""
    if not _is_dist():
        return float(x)
    bad = bool(np.isnan(x))
    t = torch.tensor([-np.inf if bad else float(x), 1.0 if bad else 0.0], dtype=torch.float64, device=_exchange_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu()
    return float('nan') if float(t[1].item()) > 0.0 else float(t[0].item())


def global_argmax(local_best: float, local_index: int, offset: int, device=None):
    """(best value, global index) over all ranks: maximum value, ties -> lowest global index; a NaN ranks above every
    number, as in numpy.argmax (BayesOpt/BayesOpt_v4.ipynb:647), so the lowest NaN index wins when there is one."""
    gi = int(local_index) + int(offset)
    if not _is_dist():
        return float(local_best), gi
    world = dist.get_world_size()
    mine = torch.tensor([float(local_best), float(gi)], dtype=torch.float64, device=_exchange_device(device))
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    vals = np.array([[float(t[0]), float(t[1])] for t in allv])     # indices < 2^53 are exact in f64
    return pick_best(vals[:, 0], vals[:, 1].astype(np.int64))


def global_argmax_payload(local_best: float, local_index: int, offset: int, payload, device=None):
    """global_argmax that also hands every rank a small vector belonging to the winner (the coordinates of the chosen
    candidate): one all-gather of (value, global index, payload...) per rank instead of a gather followed by a sum."""
    gi = int(local_index) + int(offset)
    payload = np.ascontiguousarray(payload, dtype=np.float64).reshape(-1)
    if not _is_dist():
        return float(local_best), gi, payload.copy()
    world = dist.get_world_size()
    mine = torch.from_numpy(np.concatenate(([float(local_best), float(gi)], payload))).to(_exchange_device(device))
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    rows = np.stack([t.cpu().numpy() for t in allv])                 # indices < 2^53 are exact in f64
    best, gidx = pick_best(rows[:, 0], rows[:, 1].astype(np.int64))
    owner = int(np.flatnonzero(rows[:, 1].astype(np.int64) == gidx)[0])
    return best, gidx, rows[owner, 2:].copy()


def global_first_at_least(gp, threshold, offset, fallback):
    """Lowest GLOBAL index whose acquisition value is >= threshold (the tie-tolerant argmax once the global maximum is
    known): each rank scans its block, one all-reduce(MAX) of the negated index picks the minimum."""
    li = gp.first_at_least(threshold)
    mine = float(li + int(offset)) if li >= 0 else float(2 ** 62)
    g = -allreduce_max(-mine, gp.device)
    return int(g) if g < 2 ** 62 else int(fallback)


def pick_best(values, indices):
    values = np.asarray(values, dtype=np.float64)
    indices = np.asarray(indices, dtype=np.int64)
    bad = np.isnan(values)
    if bad.any():
        return float('nan'), int(indices[bad].min())
    vmax = values.max()
    return float(vmax), int(indices[values == vmax].min())


class _DevBuf:
    """A ctx-owned device buffer exposed through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes) // 8,), 'typestr': '<f8', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def model_tensors(gp):
    """Zero-copy float64 views of the model buffers a non-fitting rank needs (W, alpha, Xts, params)."""
    sd = gp.state_desc()
    dev = torch.device('cuda', gp.device)
    out = []
    for ptr, nbytes in ((sd.W, sd.W_bytes), (sd.alpha, sd.alpha_bytes), (sd.Xts, sd.Xts_bytes), (sd.params, sd.params_bytes)):
        out.append(torch.as_tensor(_DevBuf(ptr, nbytes), device=dev))
    return out, sd


def slice_bounds(length: int, parts: int, s: int):
    """Element range [lo, hi) of slice s when `length` elements are cut into `parts` contiguous slices."""
    return length * s // parts, length * (s + 1) // parts


def scatter_allgather(t, src: int = 0):
    """Give every rank the contents rank `src` holds in the 1-D tensor t (same length everywhere), using every link once
    instead of one root fan-out: the tensor is cut into world - 1 slices; stage 1: src sends slice s to the s-th other rank
    (each of the root's links carries one slice); stage 2: the receivers exchange their slices among themselves, pairwise
    (each peer-to-peer link carries one slice each way).  Per link 2 / (world - 1) of the tensor moves, where a broadcast
    from one root pushes all of it through every link of the root, or hops it around a ring.  xGMI is point to point: on an
    8-GPU node the 8.6 GB inverse factor of a 32768-point model is 1.2 GB per link and stage."""
    world, rank = dist.get_world_size(), dist.get_rank()
    receivers = [r for r in range(world) if r != src]
    P = len(receivers)
    n = t.numel()
    ops = []
    if rank == src:
        for s, r in enumerate(receivers):
            lo, hi = slice_bounds(n, P, s)
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, t[lo:hi], r))
    else:
        lo, hi = slice_bounds(n, P, receivers.index(rank))
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, t[lo:hi], src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank == src or P == 1:
        return
    mine = receivers.index(rank)
    mlo, mhi = slice_bounds(n, P, mine)
    ops = []
    for s, q in enumerate(receivers):
        if q == rank:
            continue
        lo, hi = slice_bounds(n, P, s)
        if mhi > mlo:
            ops.append(dist.P2POp(dist.isend, t[mlo:mhi], q))
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, t[lo:hi], q))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def broadcast_model(gp, src: int = 0):
    """After gp.fit() on rank `src`: give every other rank the same model.  Shapes travel first (one small tensor), then
    the device buffers are filled in place: alpha, the scaled inputs and the scalar block by a plain broadcast (a few
    hundred kilobytes), the inverse factor W by scatter_allgather."""
    if not _is_dist():
        return
    rank = dist.get_rank()
    xdev = _exchange_device(gp.device)
    hdr = torch.zeros(9, dtype=torch.float64, device=xdev)
    if rank == src:
        sd = gp.state_desc()
        hdr = torch.tensor([sd.n, sd.d, sd.kernel, sd.variant, sd.sigmaY, sd.y_max, gp.theta1, gp.nll, sd.n_pad],
                           dtype=torch.float64, device=xdev)
    dist.broadcast(hdr, src)
    hdr = hdr.cpu()
    n, d, kernel, variant = (int(hdr[i].item()) for i in range(4))
    if variant == 4:
        raise RuntimeError('broadcast_model: a trend model that fell back to LU (K not positive definite) carries row exchanges '
                           'this hand-over does not move; use one process with MultiGP for it')
    sigmaY, y_max, n_pad = float(hdr[4].item()), float(hdr[5].item()), int(hdr[8].item())
    if rank != src:
        gp.state_alloc(n, d, kernel, variant, sigmaY, n_pad=n_pad)     # the sender's padding: it sizes every buffer below
    tensors, sd_local = model_tensors(gp)
    want = (8 * n_pad * n_pad, 8 * n_pad)
    if (sd_local.W_bytes, sd_local.alpha_bytes) != want:
        raise RuntimeError('broadcast_model: rank %d holds buffers of %r bytes, the sender\'s are %r' %
                           (rank, (sd_local.W_bytes, sd_local.alpha_bytes), want))
    W, small = tensors[0], tensors[1:]
    if dist.get_backend() == 'nccl':
        torch.cuda.synchronize(gp.device)
        for t in small:
            dist.broadcast(t, src)
        if os.environ.get('GPBO_BCAST', 'scatter') == 'root':      # one-root fan-out, for comparison runs
            dist.broadcast(W, src)
        else:
            scatter_allgather(W, src)
        torch.cuda.synchronize(gp.device)
    else:                                            # gloo: stage through host memory
        for t in small:
            h = t.cpu()
            dist.broadcast(h, src)
            if rank != src:
                t.copy_(h)
        h = W.cpu()
        scatter_allgather(h, src)
        if rank != src:
            W.copy_(h)
        torch.cuda.synchronize(gp.device)
    if rank != src:
        gp.state_commit(y_max)
        gp.theta1, gp.nll = float(hdr[6].item()), float(hdr[7].item())


class ShardedSweep:
    """posterior -> all-reduce(max mu) -> acquisition -> all-gather argmax over this rank's candidate block."""

    def __init__(self, gp, offset: int = 0):
        self.gp = gp
        self.offset = int(offset)

    def step(self, acq=0, kappa=0.01, tie_rtol=0.0):
        mu_max_local = self.gp.posterior()
        mu_max = allreduce_max(mu_max_local, self.gp.device)
        idx, best = self.gp.acquire(acq, kappa, 2, mu_max)          # incumbent given = global max(Mu)
        gbest, gidx = global_argmax(best, idx, self.offset, self.gp.device)
        if tie_rtol > 0.0 and np.isfinite(gbest):
            gidx = global_first_at_least(self.gp, gbest - tie_rtol * abs(gbest), self.offset, gidx)
        return gidx, gbest, mu_max


def allreduce_sum(vec, device=None):
    """Element-wise sum over ranks of a small float64 vector (returned as a NumPy array)."""
    v = np.ascontiguousarray(vec, dtype=np.float64)
    if not _is_dist():
        return v.copy()
    t = torch.from_numpy(v.copy()).to(_exchange_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class ShardedLoop:
    """Frozen-theta BO loop (single-factor posterior, GP/BayesOptFcn_v1.m:73-83) with the candidate rows -- and with them the
    resident V = W Ks^T -- split over ranks.  Every rank holds the same small model and applies the same border-append, so
    the model needs no exchange; per iteration the data path has two exchanges: all-reduce(MAX) of max(Mu) (one double) and
    one all-gather of (acquisition max, global index, the d coordinates of that candidate) per rank."""

    def __init__(self, gp, candidates_local, offset: int = 0, exchange: bool = True):
        """exchange=False: this rank's loop is its own (independent replicas under one launcher): nothing is exchanged."""
        self.gp = gp
        self.X = np.ascontiguousarray(candidates_local, dtype=np.float64)
        self.offset = int(offset)
        self.exchange = bool(exchange) and _is_dist()
        gp.set_candidates(self.X)
        self.mu_max = self._max(gp.posterior())

    def _max(self, v):
        return allreduce_max(v, self.gp.device) if self.exchange else v

    @classmethod
    def on_lattice(cls, gp, axes, exchange: bool = True):
        """The same loop on a candidate LATTICE (every combination of the per-axis coordinates, last axis fastest): the ranks
        take contiguous slabs of axis 0, each slab again a lattice, generated in HBM (GP.set_candidates_lattice).  With the
        RBF kernel an append then costs one small matrix product per rank instead of a pass over a resident (n, m / ranks)
        matrix, and a candidate's numbers do not depend on the number of ranks."""
        self = cls.__new__(cls)
        self.exchange = bool(exchange) and _is_dist()
        axes = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)) for a in axes]
        rank = dist.get_rank() if self.exchange else 0
        world = dist.get_world_size() if self.exchange else 1
        lo, hi = shard_range(len(axes[0]), rank, world)
        if hi <= lo:
            raise ValueError('on_lattice: axis 0 has %d points, fewer than the %d ranks' % (len(axes[0]), world))
        self.gp = gp
        self.X = None
        self.axes = [axes[0][lo:hi]] + axes[1:]
        self.offset = int(lo) * int(np.prod([len(a) for a in axes[1:]], dtype=np.int64))
        gp.set_candidates_lattice(self.axes)
        self.mu_max = self._max(gp.posterior())
        return self

    def _point(self, idx):
        """Coordinates of local candidate idx."""
        if self.X is not None:
            return self.X[idx].copy()
        x = np.empty(len(self.axes))
        r = int(idx)
        for t in range(len(self.axes) - 1, -1, -1):
            r, g = divmod(r, len(self.axes[t]))
            x[t] = self.axes[t][g]
        return x

    def propose(self, acq=0, kappa=0.01):
        """(global index, acquisition value, coordinates) of the next point; identical on every rank."""
        idx, best = self.gp.acquire(acq, kappa, 2, self.mu_max)
        if not self.exchange:
            return idx + self.offset, best, self._point(idx)
        gbest, gidx, x = global_argmax_payload(best, idx, self.offset, self._point(idx), self.gp.device)
        return gidx, gbest, x

    def observe(self, x_new, y_new):
        self.mu_max = self._max(self.gp.append(x_new, y_new))
        return self.mu_max

    def step(self, x_new, y_new, acq=0, kappa=0.01):
        """observe(x_new, y_new) then propose(acq, kappa): one loop iteration.  With a single rank the two halves share
        one library call and one host synchronisation (the incumbent max(Mu) needs no exchange); the picks are the same."""
        if self.exchange:
            self.observe(x_new, y_new)
            return self.propose(acq, kappa)
        self.mu_max, idx, best = self.gp.append_acquire(x_new, y_new, acq, kappa, 0)      # incumbent = max(Mu), on the device
        gidx = idx + self.offset
        return gidx, best, self._point(idx)


def sharded_lml(gp, xTrain, yTrain, thetas, sigmas, kernel=0, grad=True):
    """-logL (and gradient) of B hyper-parameter settings, settings split over ranks in contiguous blocks; every rank
    returns the full (nll (B,), grad (B,d) or None, status (B,)).  The settings are independent: the only exchange is
    the gather of the results."""
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    sigmas = np.ascontiguousarray(sigmas, dtype=np.float64).reshape(-1)
    B, d = thetas.shape
    rank = dist.get_rank() if _is_dist() else 0
    world = dist.get_world_size() if _is_dist() else 1
    lo, hi = shard_range(B, rank, world)
    buf = np.zeros((B, d + 2))
    if hi > lo:
        res = gp.lml_batch(xTrain, yTrain, thetas[lo:hi], sigmas[lo:hi], kernel=kernel, grad=grad)
        nll, g, stat = res if grad else (res[0], None, res[1])
        buf[lo:hi, 0] = np.where(np.isnan(nll), 0.0, nll)          # NaN (failed setting) travels as status, not through the sum
        buf[lo:hi, 1] = stat
        if grad:
            buf[lo:hi, 2:] = np.where(np.isnan(g), 0.0, g)
    buf = allreduce_sum(buf.reshape(-1), gp.device).reshape(B, d + 2)
    stat = buf[:, 1].astype(np.int32)
    nll = np.where(stat != 0, np.nan, buf[:, 0])
    g = np.where((stat != 0)[:, None], np.nan, buf[:, 2:]) if grad else None
    return nll, g, stat
