"""Device memory returns to where it was after create -> work -> close cycles of a context: nothing the library allocates
(model buffers, the likelihood sweep's work contexts, lattice factor tables, event pools) and nothing the runtime allocates
on its behalf outlives gpbo_destroy."""
import numpy as np
import pytest
import torch

import bayesian_optimization_amd as bo
from bayesian_optimization_amd import GP, MultiGP

pytestmark = pytest.mark.gpu


def _free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _cycle(rng, n, d):
    X = rng.random((n, d))
    y = np.sin(3 * X.sum(1))
    Xs = rng.random((5000, d))
    th = np.full(d, 0.5)
    gp = GP(0)
    gp.fit(X, y, th, 0.1)
    gp.predict_ei(Xs, 0.01)
    gp.factor()
    gp.lml_batch(X, y, np.tile(th, (4, 1)) * np.linspace(0.8, 1.2, 4)[:, None], np.full(4, 0.1), grad=True)
    gp.close()
    mg = MultiGP([0, 0])
    mg.reserve_appends(3)
    mg.fit(X[:30], y[:30], th, 0.05, variant=bo.VARIANT_SINGLE)
    axes = [np.linspace(0, 1, 9)] * d
    mg.set_candidates_lattice(axes)
    mu_max = mg.posterior()
    j, _ = mg.acquire(bo.ACQ_EI, 0.01, bo.INCUMBENT_GIVEN, mu_max)
    for k in range(3):
        mg.append_acquire(X[40 + k], float(y[40 + k]), bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU)
    mg.read()
    mg.close()


def test_context_lifetimes_leave_no_device_memory_behind():
    """n = 600 (640 x 640 matrices, 3.1 MiB each): a single matrix lost per context lifetime would show as 60 MiB over the
    cycles below; the allowance covers the runtime's own one-off growth (code objects, queues)."""
    rng = np.random.default_rng(0)
    for _ in range(3):          # (the runtime grows once, by 192 MiB, on the second lifetime in a fresh process: queues of the worker threads)
        _cycle(rng, 600, 3)
    f0 = _free()
    for _ in range(20):
        _cycle(rng, 600, 3)
    assert (f0 - _free()) / 2 ** 20 <= 8.0


def test_lu_route_of_the_trend_flavour_leaves_no_device_memory_behind():
    """300 of the notebook's 3-D points with a long length scale: Cholesky meets a non-positive pivot, the LU route answers
    (factors, row exchanges, per-chunk solves with a strided device-to-device copy in front).  The same lifetime check."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gprnb_3d_n800.npz'))
    xT, yT, xS = g['xT'][:300], g['yT'][:300], g['xS']

    def cycle():
        gp = GP(0)
        gp.set_chunk(256)                       # several chunks, the last one partial
        gp.kriging_fit(xT, yT, 2.0, 1)
        with pytest.raises(ValueError):
            gp.factor()                         # the LU route ran: no Cholesky factor to hand out
        gp.set_candidates(xS[:1000])
        gp.posterior()
        gp.read()
        gp.close()

    for _ in range(3):
        cycle()
    f0 = _free()
    for _ in range(15):
        cycle()
    assert (f0 - _free()) / 2 ** 20 <= 8.0
