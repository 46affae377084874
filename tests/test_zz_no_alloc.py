"""GPU: gpbo_acquire_static (the entry point that allocates nothing: reduction partials in a file-scope device array)
against gpbo_acquire on the same posterior, through the C ABI.  Bit-exact: the same kernels on the same numbers.
Two calls in a row on one GPU, the second with a smaller input (the shared scratch keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

import bayesian_optimization_amd as bo
from bayesian_optimization_amd import _lib
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def _static_fn():
    return _lib.load().gpbo_acquire_static


def _call(fn, mu_t, var_t, out_t, m, acq, kappa, mode, inc, stream=None, res_t=None):
    idx, best, mm = C.c_int64(-1), C.c_double(), C.c_double()
    st = fn(0, stream, mu_t.data_ptr(), var_t.data_ptr(), m, acq, kappa, mode, inc, out_t.data_ptr(),
            res_t.data_ptr() if res_t is not None else None, C.byref(idx), C.byref(best), C.byref(mm))
    return st, int(idx.value), best.value, mm.value


def test_static_acquire_equals_acquire_twice_in_a_row(gp0):
    import torch
    fn = _static_fn()
    n, d = 300, 3
    X = o.splitmix64_uniform(11, n * d).reshape(n, d)
    y = np.sin(3 * X.sum(1))
    gp0.fit(X, y, np.array([0.4, 0.5, 0.6]), 0.05)
    for m, acq, kappa, mode in ((300001, bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU), (4097, bo.ACQ_EI, 0.01, bo.INCUMBENT_MAX_MU),
                                (1000, bo.ACQ_UCB, 2.0, bo.INCUMBENT_MAX_MU), (77, bo.ACQ_PI, 0.0, bo.INCUMBENT_GIVEN)):
        Xs = o.splitmix64_uniform(1000 + m, m * d).reshape(m, d)
        gp0.set_candidates(Xs)
        mu_max = gp0.posterior()
        inc = 0.3 if mode == bo.INCUMBENT_GIVEN else 0.0
        idx, best = gp0.acquire(acq, kappa, mode, inc)
        mu, var, ref = gp0.read(acq=True)
        mu_t = torch.from_numpy(mu).cuda()
        var_t = torch.from_numpy(var).cuda()
        out_t = torch.full((m + 8,), -7.0, dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        st, i2, b2, mm2 = _call(fn, mu_t, var_t, out_t, m, acq, kappa, mode, inc)
        assert st == 0
        got = out_t.cpu().numpy()
        assert np.array_equal(got[:m], ref) and np.all(got[m:] == -7.0)      # bit-exact, nothing written past m
        assert (i2, b2, mm2) == (idx, best, mu_max)
        # launch-only form (no host out-pointer: no copy, no synchronisation) on a stream of the caller's: the four results
        # land in caller-owned device memory
        s2 = torch.cuda.Stream()
        res_t = torch.zeros(4, dtype=torch.float64, device='cuda')
        out_t.fill_(-7.0)
        torch.cuda.synchronize()
        assert fn(0, s2.cuda_stream, mu_t.data_ptr(), var_t.data_ptr(), m, acq, kappa, mode, inc, out_t.data_ptr(), res_t.data_ptr(),
                  None, None, None) == 0
        s2.synchronize()
        res = res_t.cpu().numpy()
        assert np.array_equal(out_t.cpu().numpy()[:m], ref)
        assert res[1] == mu_max and res[2] == best and int(res[3:4].view(np.int64)[0]) == idx
        assert res[0] == (inc if mode == bo.INCUMBENT_GIVEN else mu_max)
    # sizes beyond the stated maximum and modes that need a model are rejected before anything is launched
    assert fn(0, None, mu_t.data_ptr(), var_t.data_ptr(), (1 << 28) + 1, 0, 0.01, 0, 0.0, out_t.data_ptr(), None, None, None, None) == 2
    assert fn(0, None, mu_t.data_ptr(), var_t.data_ptr(), 10, 0, 0.01, bo.INCUMBENT_MAX_Y, 0.0, out_t.data_ptr(), None, None, None, None) == 2
    assert fn(0, None, None, var_t.data_ptr(), 10, 0, 0.01, 0, 0.0, out_t.data_ptr(), None, None, None, None) == 2
