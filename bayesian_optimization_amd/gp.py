"""Host-side object API over the C ABI: one ``GP`` = one GPU context.

fit -> set_candidates -> posterior -> acquire mirrors the body of the reference's BO
loop (BayesOpt/BayesOpt_v4.ipynb:642-647); the notebook-signature wrappers live in
``notebook_api.py``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class GpboError(RuntimeError):
    pass


def _f64(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError('expected a %d-D array, got shape %s' % (ndim, a.shape))
    return a


class GP:
    def __init__(self, device: int = 0, _borrowed=None):
        self._lib = L.load()
        if _borrowed is not None:      # a context owned by a MultiGP handle: used, never destroyed, from here
            self._h, self._owned = C.c_void_p(_borrowed), False
        else:
            h = C.c_void_p()
            st = self._lib.gpbo_create(int(device), C.byref(h))
            if st != L.OK or not h.value:
                raise GpboError('gpbo_create(device=%d) failed with status %d (no usable GPU?)' % (device, st))
            self._h, self._owned = h, True
        self.device = int(device)
        self.n = self.d = self.m = 0
        self._xs_rows = 0        # candidate rows resident on the device (m is the length of the current posterior)
        self.theta1 = self.nll = float('nan')
        # argument and result slots of the per-iteration calls (append, append_acquire), made once: building the ctypes objects
        # anew on every call costs 4 us of a 240 us loop iteration
        self._xbuf = (C.c_double * 32)()
        self._o = [C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()]
        self._r = [C.byref(v) for v in self._o]

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            if self._owned:
                self._lib.gpbo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st == L.OK:
            return
        msg = self._lib.gpbo_last_error(self._h)
        msg = msg.decode() if msg else ''
        if st == L.ERR_NOT_PD:
            # what np.linalg.cholesky raises at BayesOpt_v4.ipynb:142
            raise np.linalg.LinAlgError(msg or 'Matrix is not positive definite')
        if st == L.ERR_BAD_ARG:
            raise ValueError(msg)
        raise GpboError('HIP error: ' + msg)

    @property
    def stream(self):
        return self._lib.gpbo_stream(self._h)

    def timings(self):
        t = L.Timings()
        self._check(self._lib.gpbo_last_timings(self._h, C.byref(t)))
        return t.as_dict()

    def set_timing(self, enabled: bool):
        """Per-phase event timing (what timings() reads) on or off; off saves the events' cost in tight loops of step calls."""
        self._check(self._lib.gpbo_set_timing(self._h, 1 if enabled else 0))

    def set_chunk(self, chunk: int):
        self._check(self._lib.gpbo_set_chunk(self._h, int(chunk)))

    # ------------------------------------------------------------------ model
    def fit(self, xTrain, yTrain, theta, sigmaY, kernel=L.KERNEL_RBF, variant=L.VARIANT_V4):
        X = _f64(xTrain, 2)
        y = _f64(np.asarray(yTrain).reshape(-1), 1)
        th = _f64(np.asarray(theta).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError('xTrain has %d rows, yTrain %d' % (X.shape[0], y.shape[0]))
        if th.shape[0] != X.shape[1]:
            raise ValueError('theta has %d entries, xTrain %d columns' % (th.shape[0], X.shape[1]))
        t1, nll = C.c_double(), C.c_double()
        try:
            self._check(self._lib.gpbo_fit(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1], L.dptr(th),
                                           float(sigmaY), int(kernel), int(variant), C.byref(t1), C.byref(nll)))
        finally:
            self._sync_shape()   # candidates stay resident across refits unless the column count changes; also after a failed fit
        self.theta1, self.nll = t1.value, nll.value
        return self

    def _sync_shape(self):
        """n, d, m and the resident candidate rows as the library holds them (it replaces them before a fit can fail)."""
        n, d, m, rows = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        self._lib.gpbo_shape(self._h, C.byref(n), C.byref(d), C.byref(m), C.byref(rows))
        self.n, self.d, self.m, self._xs_rows = int(n.value), int(d.value), int(m.value), int(rows.value)

    def set_candidates(self, xTest):
        Xs = _f64(xTest, 2)
        if Xs.shape[1] != self.d:
            raise ValueError('xTest has %d columns, the model %d' % (Xs.shape[1], self.d))
        self._check(self._lib.gpbo_set_candidates(self._h, Xs.ctypes.data_as(C.c_void_p), Xs.shape[0], 0))
        self.m = self._xs_rows = Xs.shape[0]
        return self

    def set_candidates_device(self, dev_ptr: int, m: int):
        """Candidates already in HBM (e.g. a torch tensor's data_ptr()); copied device-to-device."""
        self._check(self._lib.gpbo_set_candidates(self._h, C.c_void_p(int(dev_ptr)), int(m), 1))
        self.m = self._xs_rows = int(m)
        return self

    def set_candidates_uniform(self, seed: int, row_offset: int, m: int):
        """Rows [row_offset, row_offset + m) of synth.uniform's candidate stream, generated in HBM."""
        self._check(self._lib.gpbo_set_candidates_uniform(self._h, int(seed), int(row_offset), int(m)))
        self.m = self._xs_rows = int(m)
        return self

    def set_candidates_lattice(self, axes):
        """Candidates = every combination of the per-axis coordinates in ``axes`` (one 1-D array per input dimension),
        lexicographic with the last axis fastest -- np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape(-1, d) --
        generated in HBM.  With the RBF kernel and reserve_appends(), append / append_acquire then update the posterior
        through one small matrix product instead of a pass over a resident (n, m) matrix."""
        ax = [_f64(np.asarray(a).reshape(-1), 1) for a in axes]
        if len(ax) != self.d:
            raise ValueError('%d axes given, the model has %d input columns' % (len(ax), self.d))
        flat = _f64(np.concatenate(ax), 1)
        sizes = (C.c_int32 * len(ax))(*[a.shape[0] for a in ax])
        try:
            self._check(self._lib.gpbo_set_candidates_lattice(self._h, L.dptr(flat), sizes, len(ax)))
        finally:
            self._sync_shape()
        return self

    def posterior(self):
        """Mean/variance for the resident candidates (kept in HBM); returns max(mu)."""
        mm = C.c_double()
        self._check(self._lib.gpbo_posterior(self._h, C.byref(mm)))
        self.m = self._xs_rows   # (an uploaded posterior of another length is replaced)
        return mm.value

    def acquire(self, acq=L.ACQ_EI, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0, tie_rtol=0.0):
        """Returns (argmax index, acquisition maximum).  tie_rtol = 0: np.argmax semantics (first occurrence of the exact
        maximum).  tie_rtol > 0: the lowest index whose value is within tie_rtol * |max| of the maximum, a choice that does not
        depend on the last bits of values that are mathematically equal (mirror-image candidates on a lattice)."""
        idx, best = C.c_int64(), C.c_double()
        self._check(self._lib.gpbo_acquire(self._h, int(acq), float(kappa), int(incumbent_mode), float(incumbent),
                                           C.byref(idx), C.byref(best)))
        if tie_rtol > 0.0 and np.isfinite(best.value):
            first = self.first_at_least(best.value - float(tie_rtol) * abs(best.value))
            if first >= 0:
                return first, best.value
        return int(idx.value), best.value

    def first_at_least(self, threshold):
        """Lowest index whose acquisition value is >= threshold (-1: none)."""
        idx = C.c_int64()
        self._check(self._lib.gpbo_first_at_least(self._h, float(threshold), C.byref(idx)))
        return int(idx.value)

    def read(self, mu=True, var=True, acq=False, offset=0, count=None):
        count = self.m - offset if count is None else count
        out = []
        bufs = []
        for want in (mu, var, acq):
            bufs.append(np.empty(count) if want else None)
        self._check(self._lib.gpbo_read(self._h, *[L.dptr(b) if b is not None else None for b in bufs], int(offset), int(count)))
        out = [b for b in bufs if b is not None]
        return out[0] if len(out) == 1 else tuple(out)

    def predict_ei(self, xTest, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0):
        """One-call path over host buffers: (mu, var, ei, argmax, ei_max, mu_max)."""
        Xs = _f64(xTest, 2)
        if Xs.shape[1] != self.d:
            raise ValueError('xTest has %d columns, the model %d' % (Xs.shape[1], self.d))
        m = Xs.shape[0]
        mu, var, ei = np.empty(m), np.empty(m), np.empty(m)
        idx, eim, mum = C.c_int64(), C.c_double(), C.c_double()
        self._check(self._lib.gpbo_predict_ei(self._h, L.dptr(Xs), m, float(kappa), int(incumbent_mode), float(incumbent),
                                              L.dptr(mu), L.dptr(var), L.dptr(ei), C.byref(idx), C.byref(eim), C.byref(mum)))
        self.m = self._xs_rows = m
        return mu, var, ei, int(idx.value), eim.value, mum.value

    def kriging_fit(self, xT, yT, theta, deg):
        """gpr_poly (Gaussian_Process_Regression.ipynb:197-203): returns (beta (p,), cov)."""
        X = _f64(xT, 2)
        y = _f64(np.asarray(yT).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError('xT has %d rows, yT %d' % (X.shape[0], y.shape[0]))
        beta = np.zeros(10)
        nb, cov = C.c_int32(), C.c_double()
        try:
            self._check(self._lib.gpbo_kriging_fit(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1],
                                                   float(np.squeeze(theta)), int(deg), L.dptr(beta), C.byref(nb), C.byref(cov)))
        finally:
            self._sync_shape()   # as in fit(): resident candidates survive a refit with the same column count
        self.theta1, self.nll = cov.value, float('nan')
        return beta[:nb.value].copy(), cov.value

    def kriging_cross_scale(self, xT, theta_r):
        """y_kriging / mse_kriging: correlation vectors with length scale theta_r = sqrt(w0) against the fitted K."""
        X = _f64(xT, 2)
        if X.shape != (self.n, self.d):
            raise ValueError('xT must be the (%d, %d) training inputs of the fit' % (self.n, self.d))
        self._check(self._lib.gpbo_kriging_cross_scale(self._h, L.dptr(X), float(theta_r)))

    def kriging_lml(self, xT, yT, thetas, deg, epsilon):
        """gpr_mle (Gaussian_Process_Regression.ipynb:255-263) for every theta in thetas: (values (B,), status (B,))."""
        X = _f64(xT, 2)
        y = _f64(np.asarray(yT).reshape(-1), 1)
        th = _f64(np.asarray(thetas).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError('xT has %d rows, yT %d' % (X.shape[0], y.shape[0]))
        out = np.empty(th.shape[0])
        stat = np.zeros(th.shape[0], dtype=np.int32)
        self._check(self._lib.gpbo_kriging_lml(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1], L.dptr(th), th.shape[0],
                                               int(deg), float(epsilon), L.dptr(out), stat.ctypes.data_as(C.POINTER(C.c_int32))))
        return out, stat

    def set_posterior(self, mu, var):
        """Upload a (mu, diag(Cov)) pair so acquire() can evaluate an acquisition on it."""
        mu = _f64(np.asarray(mu).reshape(-1), 1)
        var = _f64(np.asarray(var).reshape(-1), 1)
        if mu.shape != var.shape:
            raise ValueError('mu and var must have the same length')
        self._check(self._lib.gpbo_set_posterior(self._h, L.dptr(mu), L.dptr(var), mu.shape[0]))
        self.m = mu.shape[0]

    def kernel_matrix(self, U, V, theta, kernel=L.KERNEL_RBF):
        U, V = _f64(U, 2), _f64(V, 2)
        th = _f64(np.asarray(theta).reshape(-1), 1)
        if U.shape[1] != V.shape[1] or th.shape[0] != U.shape[1]:
            raise ValueError('U, V and theta disagree on the number of columns')
        K = np.empty((U.shape[0], V.shape[0]))
        self._check(self._lib.gpbo_kernel_matrix(self._h, L.dptr(U), U.shape[0], L.dptr(V), V.shape[0], U.shape[1], L.dptr(th),
                                                 int(kernel), L.dptr(K)))
        return K

    def kernel_from_distances(self, Daa, theta, kernel=L.KERNEL_RBF):
        """kernfcn_v2 on a list of d per-dimension |difference| matrices, each (p, q)."""
        D = _f64(np.stack([np.asarray(a, dtype=np.float64) for a in Daa]), 3)
        th = _f64(np.asarray(theta).reshape(-1)[:D.shape[0]], 1)
        K = np.empty(D.shape[1:])
        self._check(self._lib.gpbo_kernel_from_distances(self._h, L.dptr(D), D.shape[0], D.shape[1], D.shape[2], L.dptr(th),
                                                         int(kernel), L.dptr(K)))
        return K

    def reserve_appends(self, max_appends: int):
        """Room for appended points; call before fit(variant=VARIANT_SINGLE)."""
        self._check(self._lib.gpbo_reserve_appends(self._h, int(max_appends)))

    def _point_arg(self, x_new):
        """x_new (d numbers) in the instance's argument buffer."""
        x = np.asarray(x_new, dtype=np.float64).reshape(-1)
        if x.shape[0] != self.d:
            raise ValueError('x_new has %d entries, the model %d columns' % (x.shape[0], self.d))
        self._xbuf[:self.d] = x.tolist()
        return self._xbuf

    def append(self, x_new, y_new):
        o, r = self._o, self._r
        self._check(self._lib.gpbo_append(self._h, self._point_arg(x_new), float(y_new), r[0]))
        self._refresh_scalars()
        return o[0].value

    def append_acquire(self, x_new, y_new, acq=L.ACQ_EI, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0):
        """append(x_new, y_new) + acquire(...) over the updated posterior in one call: (max(mu), argmax index, acquisition maximum)."""
        o, r = self._o, self._r
        self._check(self._lib.gpbo_append_acquire(self._h, self._point_arg(x_new), float(y_new), int(acq), float(kappa), int(incumbent_mode),
                                                  float(incumbent), r[0], r[1], r[2]))
        self._refresh_scalars()
        return o[0].value, int(o[1].value), o[2].value

    def _refresh_scalars(self):
        o, r = self._o, self._r
        self._check(self._lib.gpbo_model_scalars(self._h, r[3], r[4], r[5], r[6]))
        self.theta1, self.nll, self.y_max, self.n = o[3].value, o[4].value, o[5].value, int(o[6].value)

    def cond_estimate(self):
        """(max diag L / min diag L)^2 of the factor the posterior uses: a lower bound of cond_2 of the factored matrix."""
        c = C.c_double()
        self._check(self._lib.gpbo_cond_estimate(self._h, C.byref(c)))
        return c.value

    def factor(self, want_L=True, want_alpha=True):
        if want_L:
            self.factor_downloads = getattr(self, 'factor_downloads', 0) + 1      # n^2 doubles device -> host each
        Lm = np.empty((self.n, self.n)) if want_L else None
        al = np.empty(self.n) if want_alpha else None
        self._check(self._lib.gpbo_get_factor(self._h, L.dptr(Lm) if want_L else None, L.dptr(al) if want_alpha else None))
        return Lm, al

    def lml_batch(self, xTrain, yTrain, thetas, sigmas, kernel=L.KERNEL_RBF, grad=False):
        X = _f64(xTrain, 2)
        y = _f64(np.asarray(yTrain).reshape(-1), 1)
        th = _f64(thetas, 2)
        sg = _f64(np.asarray(sigmas).reshape(-1), 1)
        B = th.shape[0]
        if th.shape[1] != X.shape[1] or sg.shape[0] != B:
            raise ValueError('thetas must be (B,d) and sigmas (B,)')
        nll = np.empty(B)
        g = np.empty((B, X.shape[1])) if grad else None
        stat = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpbo_lml_batch(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1], L.dptr(th), L.dptr(sg),
                                             B, int(kernel), L.dptr(nll), L.dptr(g) if grad else None,
                                             stat.ctypes.data_as(C.POINTER(C.c_int32))))
        return (nll, g, stat) if grad else (nll, stat)

    def descent_small(self, xTrain, yTrain, W0, sigmaY, alpha, beta, norm, steprule, max_its, kernel=L.KERNEL_RBF):
        """The reference's descent on -logL for every row of W0, to the end on the device (n <= 128, d <= 16).
        norm: 0 'component', 1 'full', 2 raw gradient; steprule: 0 constant, 1 'diminishing', 2 'backtrack'.
        Returns (w_hist (nruns, max_its+1, d), c_hist (nruns, max_its+1), count (nruns,), exitflag (nruns,)); only the first
        count[r] entries of start r are meaningful."""
        X = _f64(xTrain, 2)
        y = _f64(np.asarray(yTrain).reshape(-1), 1)
        W = _f64(W0, 2)
        if W.shape[1] != X.shape[1] or y.shape[0] != X.shape[0]:
            raise ValueError('W0 must be (nruns, d) and yTrain (n,)')
        nruns, d, H = W.shape[0], W.shape[1], int(max_its) + 1
        wh = np.zeros((nruns, H, d))
        ch = np.full((nruns, H), np.nan)
        cnt = np.zeros(nruns, dtype=np.int32)
        flg = np.zeros(nruns, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._check(self._lib.gpbo_descent_small(self._h, L.dptr(X), L.dptr(y), X.shape[0], d, L.dptr(W), nruns, float(sigmaY),
                                                 int(kernel), float(alpha), float(beta), int(norm), int(steprule), int(max_its),
                                                 L.dptr(wh), L.dptr(ch), cnt.ctypes.data_as(ip), flg.ctypes.data_as(ip)))
        return wh, ch, cnt, flg

    def state_desc(self):
        sd = L.StateDesc()
        self._check(self._lib.gpbo_state_describe(self._h, C.byref(sd)))
        return sd

    def state_alloc(self, n, d, kernel, variant, sigmaY, n_pad=0):
        """n_pad: the sender's padded size (StateDesc.n_pad); 0 = round_up(n, 128)."""
        try:
            self._check(self._lib.gpbo_state_alloc(self._h, int(n), int(n_pad), int(d), int(kernel), int(variant), float(sigmaY)))
        finally:
            self._sync_shape()

    def state_commit(self, y_max):
        self._check(self._lib.gpbo_state_commit(self._h, float(y_max)))

    def selftest_gemm(self, n=256, rate=False):
        err, r = C.c_double(), C.c_double()
        self._check(self._lib.gpbo_selftest_gemm(self._h, int(n), C.byref(err), C.byref(r) if rate else None))
        return (err.value, r.value) if rate else err.value


class MultiGP:
    """Several GPUs (or several contexts on one GPU) behind the call surface of ``GP``, in one process: the candidate set is
    split into contiguous blocks, context 0 fits and hands the model over device to device, every sweep runs on all blocks
    side by side, and max(Mu) / the argmax are reduced inside the library (include/gpbo.h, gpbo_multi_*).  Results are
    bit-identical to one ``GP`` sweeping the whole set.  ``devices``: one entry per context, e.g. [0, 1, 2, 3] or [0, 0]."""

    def __init__(self, devices):
        self._lib = L.load()
        devs = [int(v) for v in devices]
        if not devs:
            raise ValueError('MultiGP needs at least one device')
        arr = (C.c_int32 * len(devs))(*devs)
        h = C.c_void_p()
        st = self._lib.gpbo_multi_create(arr, len(devs), C.byref(h))
        if st != L.OK or not h.value:
            raise GpboError('gpbo_multi_create(devices=%r) failed with status %d (no usable GPU?)' % (devs, st))
        self._h = h
        self.devices = devs
        self.device = devs[0]
        self.contexts = [GP(dv, _borrowed=self._lib.gpbo_multi_context(h, i)) for i, dv in enumerate(devs)]
        self.n = self.d = self.m = 0
        self._xs_rows = 0
        self.theta1 = self.nll = float('nan')

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            for g in self.contexts:
                g.close()
            self._lib.gpbo_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st == L.OK:
            return
        msg = self._lib.gpbo_multi_last_error(self._h)
        msg = msg.decode() if msg else ''
        if st == L.ERR_NOT_PD:
            raise np.linalg.LinAlgError(msg or 'Matrix is not positive definite')
        if st == L.ERR_BAD_ARG:
            raise ValueError(msg)
        raise GpboError('HIP error: ' + msg)

    def _sync_shape(self):
        n, d, m, rows = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        self._lib.gpbo_multi_shape(self._h, C.byref(n), C.byref(d), C.byref(m), C.byref(rows))
        self.n, self.d, self.m, self._xs_rows = int(n.value), int(d.value), int(m.value), int(rows.value)
        for g in self.contexts:
            g._sync_shape()

    def blocks(self):
        """[(lo, hi)] rows of the current posterior per context."""
        out = []
        for i in range(len(self.contexts)):
            lo, hi = C.c_int64(), C.c_int64()
            self._check(self._lib.gpbo_multi_block(self._h, i, C.byref(lo), C.byref(hi)))
            out.append((int(lo.value), int(hi.value)))
        return out

    def timings(self):
        """Per-context timing dictionaries (context 0 first)."""
        return [g.timings() for g in self.contexts]

    def set_timing(self, enabled: bool):
        self._check(self._lib.gpbo_multi_set_timing(self._h, 1 if enabled else 0))

    def set_chunk(self, chunk: int):
        self._check(self._lib.gpbo_multi_set_chunk(self._h, int(chunk)))

    def last_distribute(self):
        """(milliseconds, bytes) of the model hand-over of the last fit."""
        ms, by = C.c_double(), C.c_double()
        self._check(self._lib.gpbo_multi_last_distribute(self._h, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    # ------------------------------------------------------------------ model
    def fit(self, xTrain, yTrain, theta, sigmaY, kernel=L.KERNEL_RBF, variant=L.VARIANT_V4):
        X = _f64(xTrain, 2)
        y = _f64(np.asarray(yTrain).reshape(-1), 1)
        th = _f64(np.asarray(theta).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError('xTrain has %d rows, yTrain %d' % (X.shape[0], y.shape[0]))
        if th.shape[0] != X.shape[1]:
            raise ValueError('theta has %d entries, xTrain %d columns' % (th.shape[0], X.shape[1]))
        t1, nll = C.c_double(), C.c_double()
        try:
            self._check(self._lib.gpbo_multi_fit(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1], L.dptr(th),
                                                 float(sigmaY), int(kernel), int(variant), C.byref(t1), C.byref(nll)))
        finally:
            self._sync_shape()
        self.theta1, self.nll = t1.value, nll.value
        return self

    def kriging_fit(self, xT, yT, theta, deg):
        X = _f64(xT, 2)
        y = _f64(np.asarray(yT).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError('xT has %d rows, yT %d' % (X.shape[0], y.shape[0]))
        beta = np.zeros(10)
        nb, cov = C.c_int32(), C.c_double()
        try:
            self._check(self._lib.gpbo_multi_kriging_fit(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1],
                                                         float(np.squeeze(theta)), int(deg), L.dptr(beta), C.byref(nb), C.byref(cov)))
        finally:
            self._sync_shape()
        self.theta1, self.nll = cov.value, float('nan')
        return beta[:nb.value].copy(), cov.value

    def set_candidates(self, xTest):
        Xs = _f64(xTest, 2)
        if Xs.shape[1] != self.d:
            raise ValueError('xTest has %d columns, the model %d' % (Xs.shape[1], self.d))
        try:
            self._check(self._lib.gpbo_multi_set_candidates(self._h, Xs.ctypes.data_as(C.c_void_p), Xs.shape[0], 0))
        finally:
            self._sync_shape()
        return self

    def set_candidates_lattice(self, axes):
        ax = [_f64(np.asarray(a).reshape(-1), 1) for a in axes]
        if len(ax) != self.d:
            raise ValueError('%d axes given, the model has %d input columns' % (len(ax), self.d))
        flat = _f64(np.concatenate(ax), 1)
        sizes = (C.c_int32 * len(ax))(*[a.shape[0] for a in ax])
        try:
            self._check(self._lib.gpbo_multi_set_candidates_lattice(self._h, L.dptr(flat), sizes, len(ax)))
        finally:
            self._sync_shape()
        return self

    def set_candidates_uniform(self, seed: int, row_offset: int, m: int):
        try:
            self._check(self._lib.gpbo_multi_set_candidates_uniform(self._h, int(seed), int(row_offset), int(m)))
        finally:
            self._sync_shape()
        return self

    def posterior(self):
        mm = C.c_double()
        try:
            self._check(self._lib.gpbo_multi_posterior(self._h, C.byref(mm)))
        finally:
            self._sync_shape()
        return mm.value

    def acquire(self, acq=L.ACQ_EI, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0, tie_rtol=0.0):
        idx, best = C.c_int64(), C.c_double()
        self._check(self._lib.gpbo_multi_acquire(self._h, int(acq), float(kappa), int(incumbent_mode), float(incumbent),
                                                 C.byref(idx), C.byref(best)))
        if tie_rtol > 0.0 and np.isfinite(best.value):
            first = self.first_at_least(best.value - float(tie_rtol) * abs(best.value))
            if first >= 0:
                return first, best.value
        return int(idx.value), best.value

    def first_at_least(self, threshold):
        idx = C.c_int64()
        self._check(self._lib.gpbo_multi_first_at_least(self._h, float(threshold), C.byref(idx)))
        return int(idx.value)

    def read(self, mu=True, var=True, acq=False, offset=0, count=None):
        count = self.m - offset if count is None else count
        bufs = [np.empty(count) if want else None for want in (mu, var, acq)]
        self._check(self._lib.gpbo_multi_read(self._h, *[L.dptr(b) if b is not None else None for b in bufs], int(offset), int(count)))
        out = [b for b in bufs if b is not None]
        return out[0] if len(out) == 1 else tuple(out)

    def predict_ei(self, xTest, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0):
        Xs = _f64(xTest, 2)
        if Xs.shape[1] != self.d:
            raise ValueError('xTest has %d columns, the model %d' % (Xs.shape[1], self.d))
        m = Xs.shape[0]
        mu, var, ei = np.empty(m), np.empty(m), np.empty(m)
        idx, eim, mum = C.c_int64(), C.c_double(), C.c_double()
        try:
            self._check(self._lib.gpbo_multi_predict_ei(self._h, L.dptr(Xs), m, float(kappa), int(incumbent_mode), float(incumbent),
                                                        L.dptr(mu), L.dptr(var), L.dptr(ei), C.byref(idx), C.byref(eim), C.byref(mum)))
        finally:
            self._sync_shape()
        return mu, var, ei, int(idx.value), eim.value, mum.value

    def set_posterior(self, mu, var):
        mu = _f64(np.asarray(mu).reshape(-1), 1)
        var = _f64(np.asarray(var).reshape(-1), 1)
        if mu.shape != var.shape:
            raise ValueError('mu and var must have the same length')
        try:
            self._check(self._lib.gpbo_multi_set_posterior(self._h, L.dptr(mu), L.dptr(var), mu.shape[0]))
        finally:
            self._sync_shape()

    def reserve_appends(self, max_appends: int):
        self._check(self._lib.gpbo_multi_reserve_appends(self._h, int(max_appends)))

    def append(self, x_new, y_new):
        x = _f64(np.asarray(x_new).reshape(-1), 1)
        if x.shape[0] != self.d:
            raise ValueError('x_new has %d entries, the model %d columns' % (x.shape[0], self.d))
        mm = C.c_double()
        try:
            self._check(self._lib.gpbo_multi_append(self._h, L.dptr(x), float(y_new), C.byref(mm)))
        finally:
            self._sync_shape()
        self._refresh_scalars()
        return mm.value

    def append_acquire(self, x_new, y_new, acq=L.ACQ_EI, kappa=0.01, incumbent_mode=L.INCUMBENT_MAX_MU, incumbent=0.0):
        x = _f64(np.asarray(x_new).reshape(-1), 1)
        if x.shape[0] != self.d:
            raise ValueError('x_new has %d entries, the model %d columns' % (x.shape[0], self.d))
        mm, idx, best = C.c_double(), C.c_int64(), C.c_double()
        try:
            self._check(self._lib.gpbo_multi_append_acquire(self._h, L.dptr(x), float(y_new), int(acq), float(kappa), int(incumbent_mode),
                                                            float(incumbent), C.byref(mm), C.byref(idx), C.byref(best)))
        finally:
            self._sync_shape()
        self._refresh_scalars()
        return mm.value, int(idx.value), best.value

    def _refresh_scalars(self):
        g = self.contexts[0]
        g._refresh_scalars()
        self.theta1, self.nll, self.y_max, self.n = g.theta1, g.nll, g.y_max, g.n

    def lml_batch(self, xTrain, yTrain, thetas, sigmas, kernel=L.KERNEL_RBF, grad=False):
        X = _f64(xTrain, 2)
        y = _f64(np.asarray(yTrain).reshape(-1), 1)
        th = _f64(thetas, 2)
        sg = _f64(np.asarray(sigmas).reshape(-1), 1)
        B = th.shape[0]
        if th.shape[1] != X.shape[1] or sg.shape[0] != B:
            raise ValueError('thetas must be (B,d) and sigmas (B,)')
        nll = np.empty(B)
        g = np.empty((B, X.shape[1])) if grad else None
        stat = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpbo_multi_lml_batch(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1], L.dptr(th), L.dptr(sg),
                                                   B, int(kernel), L.dptr(nll), L.dptr(g) if grad else None,
                                                   stat.ctypes.data_as(C.POINTER(C.c_int32))))
        return (nll, g, stat) if grad else (nll, stat)

    # what only the fitting context can answer, or what needs no split
    def factor(self, want_L=True, want_alpha=True):
        if want_L:
            self.factor_downloads = getattr(self, 'factor_downloads', 0) + 1
        return self.contexts[0].factor(want_L, want_alpha)

    def cond_estimate(self):
        return self.contexts[0].cond_estimate()

    def kernel_matrix(self, U, V, theta, kernel=L.KERNEL_RBF):
        return self.contexts[0].kernel_matrix(U, V, theta, kernel)

    def kernel_from_distances(self, Daa, theta, kernel=L.KERNEL_RBF):
        return self.contexts[0].kernel_from_distances(Daa, theta, kernel)

    def descent_small(self, *args, **kwargs):
        return self.contexts[0].descent_small(*args, **kwargs)

    def kriging_lml(self, *args, **kwargs):
        return self.contexts[0].kriging_lml(*args, **kwargs)

    def kriging_cross_scale(self, xT, theta_r):
        for g in self.contexts:
            g.kriging_cross_scale(xT, theta_r)


def device_count():
    """GPUs visible to the library (0 when there is none)."""
    return int(L.load().gpbo_device_count())


def make_gp(devices=None):
    """A ``GP`` on one device or a ``MultiGP`` over several.  devices=None: the list in the environment variable
    GPBO_DEVICES (comma-separated; an id may repeat, e.g. "0,0" = two contexts on GPU 0); else, under a one-process-per-GPU
    launcher (LOCAL_RANK set), this rank's own GPU; else every visible GPU."""
    import os
    if devices is None:
        env = os.environ.get('GPBO_DEVICES')
        if env:
            devices = [int(v) for v in env.split(',')]
        elif os.environ.get('LOCAL_RANK') is not None:
            devices = [int(os.environ['LOCAL_RANK'])]
        else:
            devices = list(range(max(1, device_count())))
    devices = [int(v) for v in np.atleast_1d(devices)]
    return GP(devices[0]) if len(devices) == 1 else MultiGP(devices)
