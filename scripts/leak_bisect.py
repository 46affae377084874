"""Device-memory drift per create -> work -> close cycle, per kind of work and per training-set size.

Run on a GPU: python3 -u scripts/leak_bisect.py
Free memory is read with hipMemGetInfo through the HIP runtime the library itself is linked against, so nothing but
the library (and the runtime underneath it) allocates between two readings.
"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import bayesian_optimization_amd as bo  # noqa: E402
from bayesian_optimization_amd import GP, MultiGP  # noqa: E402

GP(0).close()  # the library loads and initialises the runtime first
hip = ctypes.CDLL("libamdhip64.so.7")  # the soname the library is linked against: the same copy


def free_bytes():
    hip.hipDeviceSynchronize()
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def drift(name, fn, cycles=12):
    fn()
    f0 = free_bytes()
    for _ in range(cycles):
        fn()
    f1 = free_bytes()
    print("%-34s %8.2f MiB per cycle" % (name, (f0 - f1) / 2**20 / cycles), flush=True)


def raw_alloc(nbytes, count):
    def fn():
        ps = []
        for _ in range(count):
            p = ctypes.c_void_p()
            rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
            assert rc == 0, rc
            ps.append(p)
        for p in ps:
            assert hip.hipFree(p) == 0
    return fn


def raw_memset(nbytes, own_stream):
    """hipMalloc -> hipMemsetAsync -> synchronize -> hipFree (-> hipStreamDestroy): what one context lifetime did to W."""
    def fn():
        st = ctypes.c_void_p()
        if own_stream:
            assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
        assert hip.hipMemsetAsync(p, 0, ctypes.c_size_t(nbytes), st) == 0
        assert hip.hipStreamSynchronize(st) == 0
        assert hip.hipFree(p) == 0
        if own_stream:
            assert hip.hipStreamDestroy(st) == 0
    return fn


def raw_copy2d(big_bytes):
    """A context's life in small: stream, one large and one small buffer, a strided device-to-host read of the small one."""
    host = (ctypes.c_double * 4)()

    def fn():
        st, p, q = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(big_bytes)) == 0
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(512)) == 0
        src = ctypes.c_void_p(p.value + 440)
        rc = hip.hipMemcpy2DAsync(host, ctypes.c_size_t(8), src, ctypes.c_size_t(8), ctypes.c_size_t(8), ctypes.c_size_t(1),
                                  ctypes.c_int(2), st)
        assert rc == 0, rc
        assert hip.hipStreamSynchronize(st) == 0
        assert hip.hipFree(p) == 0
        assert hip.hipFree(q) == 0
        assert hip.hipStreamDestroy(st) == 0
    return fn


rng = np.random.default_rng(0)
d, m = 3, 20000
Xall = rng.random((1200, d))
yall = np.sin(3 * Xall.sum(1))
Xs = rng.random((m, d))
th = np.full(d, 0.5)


def fit_only(n, variant=None):
    def fn():
        g = GP(0)
        if variant is None:
            g.fit(Xall[:n], yall[:n], th, 0.1)
        else:
            g.fit(Xall[:n], yall[:n], th, 0.1, variant=variant)
        g.close()
    return fn


def fit_twice(n):
    def fn():
        g = GP(0)
        g.fit(Xall[:n], yall[:n], th, 0.1)
        g.fit(Xall[:n], yall[:n], th * 1.1, 0.1)
        g.close()
    return fn


def lml(n, b, grad):
    def fn():
        g = GP(0)
        g.lml_batch(Xall[:n], yall[:n], np.tile(th, (b, 1)), np.full(b, 0.1), grad=grad)
        g.close()
    return fn


def lml_same_ctx(n, b):
    g = GP(0)

    def fn():
        g.lml_batch(Xall[:n], yall[:n], np.tile(th, (b, 1)), np.full(b, 0.1), grad=True)
    return fn


def notime(n):
    def fn():
        g = GP(0)
        g.set_timing(False)
        g.fit(Xall[:n], yall[:n], th, 0.1)
        g.close()
    return fn


cases = [("hipMalloc/hipFree 1.2 MiB x 13", raw_alloc(384 * 384 * 8, 13)),
         ("hipMalloc/hipFree 3 KiB x 13", raw_alloc(3072, 13)),
         ("raw memsetAsync 4 MiB, null stream", raw_memset(4 << 20, False)),
         ("raw memsetAsync 4 MiB, own stream", raw_memset(4 << 20, True)),
         ("raw 2-D read-back beside 4 MiB", raw_copy2d(4 << 20)),
         ("raw 2-D read-back beside 16 MiB", raw_copy2d(16 << 20)),
         ("create/close", lambda: GP(0).close())]
for n in (100, 128, 129, 300, 600, 1200):
    cases.append(("fit n=%d" % n, fit_only(n)))
cases.append(("fit n=300 single", fit_only(300, bo.VARIANT_SINGLE)))
cases.append(("fit n=300 twice", fit_twice(300)))
if hasattr(GP, "set_timing"):
    cases.append(("fit n=300 timing off", notime(300)))
for n, b, gr in ((300, 1, False), (300, 1, True), (300, 4, True), (300, 16, True), (600, 4, True)):
    cases.append(("lml n=%d b=%d grad=%d" % (n, b, gr), lml(n, b, gr)))
cases.append(("lml n=300 b=4, one context", lml_same_ctx(300, 4)))


def fit_factor():
    g = GP(0)
    g.fit(Xall[:300], yall[:300], th, 0.1)
    g.factor()
    g.close()


def kriging(theta):
    def fn():
        g = GP(0)
        g.kriging_fit(Xall[:300] * 4, yall[:300], theta, 1)   # theta = 8: not positive definite, the LU route answers
        g.set_candidates(Xs * 4)
        g.posterior()
        g.read()
        g.close()
    return fn


def loop_cycle(lattice):
    def fn():
        g = GP(0)
        g.reserve_appends(6)
        g.fit(Xall[:30], yall[:30], th, 0.05, variant=bo.VARIANT_SINGLE)
        if lattice:
            g.set_candidates_lattice([np.linspace(0, 1, 24)] * d)
        else:
            g.set_candidates(Xs)
        mu = g.posterior()
        for k in range(5):
            g.append(Xall[600 + k], float(yall[600 + k]))
        g.read()
        g.close()
    return fn


cases += [("fit n=300 + factor()", fit_factor), ("kriging theta=1 (Cholesky)", kriging(1.0)), ("kriging theta=8 (LU)", kriging(8.0)),
          ("5 appends, streaming", loop_cycle(False)), ("5 appends, lattice 24^3", loop_cycle(True)),
          ("MultiGP x3 fit+posterior", lambda: multi_cycle())]


def multi_cycle():
    mg = MultiGP([0, 0, 0])
    mg.fit(Xall[:300], yall[:300], th, 0.05)
    mg.set_candidates(Xs)
    mg.posterior()
    mg.close()
for name, fn in cases:
    drift(name, fn)
