"""Multi-GPU sharding of the candidate sweep: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in CPU tests).

Candidates are split into contiguous blocks (global index = block offset + local index, so the
first-occurrence argmax survives sharding).  The data path has exactly two exchange steps:
  1. all-reduce(MAX) of one double: the EI incumbent max(Mu) (BayesOpt/BayesOpt_v4.ipynb:193);
  2. all-gather of (acquisition max, global index) per rank, reduced to max value / lowest index.
The fitted model (inverse factor W, alpha, scaled inputs, scalars) is computed on rank `src`
only and broadcast once per fit.
"""
from __future__ import annotations

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
    if not _is_dist():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_exchange_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def global_argmax(local_best: float, local_index: int, offset: int, device=None):
    """(best value, global index) over all ranks: maximum value, ties -> lowest global index.
    NaN local values never win (numpy.argmax would return the first NaN; the sweep produces none)."""
    gi = int(local_index) + int(offset)
    if not _is_dist():
        return float(local_best), gi
    world = dist.get_world_size()
    mine = torch.tensor([float(local_best), float(gi)], dtype=torch.float64, device=_exchange_device(device))
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    vals = np.array([[float(t[0]), float(t[1])] for t in allv])     # indices < 2^53 are exact in f64
    return pick_best(vals[:, 0], vals[:, 1].astype(np.int64))


def pick_best(values, indices):
    values = np.asarray(values, dtype=np.float64)
    indices = np.asarray(indices, dtype=np.int64)
    ok = ~np.isnan(values)
    if not ok.any():
        return float('nan'), int(indices.min())
    vmax = values[ok].max()
    return float(vmax), int(indices[ok & (values == vmax)].min())


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


def broadcast_model(gp, src: int = 0):
    """After gp.fit() on rank `src`: give every other rank the same model.  Shapes travel first
    (one small tensor), then the four device buffers are broadcast in place over RCCL."""
    if not _is_dist():
        return
    rank = dist.get_rank()
    xdev = _exchange_device(gp.device)
    hdr = torch.zeros(8, dtype=torch.float64, device=xdev)
    if rank == src:
        sd = gp.state_desc()
        hdr = torch.tensor([sd.n, sd.d, sd.kernel, sd.variant, sd.sigmaY, sd.y_max, gp.theta1, gp.nll],
                           dtype=torch.float64, device=xdev)
    dist.broadcast(hdr, src)
    n, d, kernel, variant = (int(hdr[i].item()) for i in range(4))
    sigmaY, y_max = float(hdr[4].item()), float(hdr[5].item())
    if rank != src:
        gp.state_alloc(n, d, kernel, variant, sigmaY)
    tensors, _ = model_tensors(gp)
    if dist.get_backend() == 'nccl':
        torch.cuda.synchronize(gp.device)
        for t in tensors:
            dist.broadcast(t, src)
        torch.cuda.synchronize(gp.device)
    else:                                            # gloo: stage through host memory
        for t in tensors:
            h = t.cpu()
            dist.broadcast(h, src)
            if rank != src:
                t.copy_(h)
        torch.cuda.synchronize(gp.device)
    if rank != src:
        gp.state_commit(y_max)
        gp.theta1, gp.nll = float(hdr[6].item()), float(hdr[7].item())


class ShardedSweep:
    """posterior -> all-reduce(max mu) -> acquisition -> all-gather argmax over this rank's candidate block."""

    def __init__(self, gp, offset: int = 0):
        self.gp = gp
        self.offset = int(offset)

    def step(self, acq=0, kappa=0.01):
        mu_max_local = self.gp.posterior()
        mu_max = allreduce_max(mu_max_local, self.gp.device)
        idx, best = self.gp.acquire(acq, kappa, 2, mu_max)          # incumbent given = global max(Mu)
        gbest, gidx = global_argmax(best, idx, self.offset, self.gp.device)
        return gidx, gbest, mu_max
