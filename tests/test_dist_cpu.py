"""CPU, world_size 2, gloo: the two exchange steps of the sharded sweep (all-reduce of max(Mu),
all-gather + first-index argmax) and the block partition, with stand-in local results."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from bayesian_optimization_amd import dist as gdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        m_total = 1001
        lo, hi = gdist.shard_range(m_total, rank, world)
        # a global "EI" array with an exact tie between the two blocks: the lower global index must win
        ei = np.sin(np.arange(m_total) * 0.37)
        ei[[100, 900]] = 5.0
        mu = np.cos(np.arange(m_total) * 0.11)
        mu_max = gdist.allreduce_max(float(mu[lo:hi].max()))
        li = int(np.argmax(ei[lo:hi]))
        best, gi = gdist.global_argmax(float(ei[lo:hi][li]), li, lo)
        # the same exchange carrying the winner's coordinates: every rank must receive those of global index 100
        coords = np.stack([np.arange(m_total) * 0.5, -np.arange(m_total, dtype=float)], axis=1)
        best2, gi2, x = gdist.global_argmax_payload(float(ei[lo:hi][li]), li, lo, coords[lo:hi][li])
        assert (best2, gi2) == (best, gi) and np.array_equal(x, coords[gi])
        q.put((rank, lo, hi, mu_max, best, gi))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m_total = 1001
    mu = np.cos(np.arange(m_total) * 0.11)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == m_total      # contiguous cover
    for r in res:
        assert r[3] == mu.max() and r[4] == 5.0 and r[5] == 100                     # same answer on every rank


def test_shard_range_covers_everything_once():
    for m in (1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            edges = [gdist.shard_range(m, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_pick_best_is_first_occurrence_and_follows_numpy_on_nan():
    assert gdist.pick_best([1.0, 3.0, 3.0, 2.0], [40, 30, 20, 10]) == (3.0, 20)
    # np.argmax returns the first NaN when there is one: the lowest NaN index wins over any number
    v, i = gdist.pick_best([1.0, float('nan'), 3.0, float('nan')], [40, 30, 20, 10])
    assert np.isnan(v) and i == 10
    v, i = gdist.pick_best([float('nan')], [5])
    assert np.isnan(v) and i == 5


class _FakeGP:
    """Stands in for a GPU context: -logL of a setting is a fixed function of the setting; one setting 'fails'."""
    device = None

    def lml_batch(self, X, y, thetas, sigmas, kernel=0, grad=False):
        nll = thetas.sum(1) + sigmas
        stat = (sigmas > 0.9).astype(np.int32)
        nll = np.where(stat != 0, np.nan, nll)
        g = np.where((stat != 0)[:, None], np.nan, 2.0 * thetas)
        return (nll, g, stat) if grad else (nll, stat)


def _lml_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        B, d = 7, 3
        thetas = np.arange(B * d, dtype=np.float64).reshape(B, d) / 10
        sigmas = np.linspace(0.1, 1.0, B)
        nll, g, stat = gdist.sharded_lml(_FakeGP(), None, None, thetas, sigmas, grad=True)
        q.put((rank, nll, g, stat))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_lml_gathers_every_setting():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_lml_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, d = 7, 3
    thetas = np.arange(B * d, dtype=np.float64).reshape(B, d) / 10
    sigmas = np.linspace(0.1, 1.0, B)
    nll, g, stat = _FakeGP().lml_batch(None, None, thetas, sigmas, grad=True)
    for r in res:
        assert np.array_equal(r[1], nll, equal_nan=True) and np.array_equal(r[2], g, equal_nan=True) and np.array_equal(r[3], stat)


def _scatter_worker(rank, world, port, q, src, length):
    import torch
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ref = torch.arange(length, dtype=torch.float64) * 0.5 + 1.0
        t = ref.clone() if rank == src else torch.full((length,), -1.0, dtype=torch.float64)
        gdist.scatter_allgather(t, src)
        # max over ranks with numpy's NaN rule: rank 1 contributes a NaN in the second call only
        a = gdist.allreduce_max(float(rank))
        b = gdist.allreduce_max(float('nan') if rank == 1 else 7.0)
        q.put((rank, bool(torch.equal(t, ref)), a, b))
    finally:
        dist.destroy_process_group()


def test_scatter_allgather_hands_every_rank_the_whole_tensor():
    """The inverse factor goes out in world - 1 slices (one per link of the root) and is completed among the receivers;
    lengths that do not divide, fewer elements than slices, and a root other than rank 0 included."""
    ctx = mp.get_context('spawn')
    for world, src, length in ((2, 0, 1000), (3, 0, 1001), (4, 2, 10), (4, 0, 2)):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q, src, length)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for r in res:
            assert r[1], (world, src, length, r[0])
            assert r[2] == world - 1 and np.isnan(r[3])


def test_slice_bounds_tile_the_range():
    for n in (0, 1, 5, 1 << 20):
        for parts in (1, 2, 3, 7):
            edges = [gdist.slice_bounds(n, parts, s) for s in range(parts)]
            assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(parts - 1))


class _LatticeStandin:
    """A context whose 'posterior' and 'acquisition' are fixed functions of a candidate's COORDINATES, so that the loop's
    host logic on lattice slabs (offsets, decoding an index into coordinates, the exchanges) can be checked without a GPU."""
    device = None

    def set_candidates_lattice(self, axes):
        self.X = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, len(axes))
        self.m = len(self.X)
        self.seen = []

    def _score(self):
        s = np.sin(7 * self.X[:, 0] + 3 * self.X[:, 1]) + np.cos(5 * self.X[:, 2])
        for x in self.seen:
            s = s - 3.0 * np.exp(-50 * np.sum((self.X - x) ** 2, 1))
        return s

    def posterior(self):
        return float(self._score().max())

    def acquire(self, acq, kappa, mode, incumbent):
        s = self._score()
        i = int(np.argmax(s))
        return i, float(s[i])

    def append(self, x, y):
        self.seen.append(np.asarray(x, dtype=np.float64))
        return self.posterior()

    def append_acquire(self, x, y, acq, kappa, mode):      # what a single rank calls: both halves in one library call
        mu_max = self.append(x, y)
        i, v = self.acquire(acq, kappa, mode, mu_max)
        return mu_max, i, v


def _lattice_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        axes = [np.linspace(0, 1, 7), np.linspace(0, 1, 5), np.linspace(0, 1, 4)]
        loop = gdist.ShardedLoop.on_lattice(_LatticeStandin(), axes)
        picks = []
        gidx, best, x = loop.propose()
        for _ in range(6):
            picks.append((int(gidx), float(best), tuple(x)))
            gidx, best, x = loop.step(x, 0.0)
        q.put((rank, loop.offset, picks))
    finally:
        dist.destroy_process_group()


def test_lattice_loop_on_slabs_of_axis_0_matches_one_process():
    """ShardedLoop.on_lattice on 3 ranks (7 axis-0 points: slabs of 3, 2, 2): offsets are whole slabs, every rank reports the
    same global index, value and coordinates, and the sequence equals the one-process loop's."""
    world, port = 3, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_lattice_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 3 * 20, 5 * 20]
    assert res[0][2] == res[1][2] == res[2][2]
    axes = [np.linspace(0, 1, 7), np.linspace(0, 1, 5), np.linspace(0, 1, 4)]
    one = gdist.ShardedLoop.on_lattice(_LatticeStandin(), axes)
    grid = np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, 3)
    gidx, best, x = one.propose()
    for k in range(6):
        assert (int(gidx), float(best), tuple(x)) == res[0][2][k] and np.array_equal(x, grid[gidx])
        gidx, best, x = one.step(x, 0.0)
