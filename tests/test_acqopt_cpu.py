"""CPU: the Nelder-Mead restatement behind acqopt.fminsearch.  MATLAB cannot run here; the restatement is pinned by the
one known answer its documentation prints (Rosenbrock's function from [-1.2, 1] with default options), and the batched
version's host logic is checked against the sequential one on stand-in contexts."""
import numpy as np

from bayesian_optimization_amd import acqopt
from oracle import gp_oracle as o


def rosenbrock(x):
    return 100.0 * (x[1] - x[0] ** 2) ** 2 + (1.0 - x[0]) ** 2


def test_restatement_reproduces_the_documented_fminsearch_example():
    """fminsearch(@(x) 100*(x(2)-x(1)^2)^2 + (1-x(1))^2, [-1.2, 1]) as printed in the MATLAB documentation:
    x = 1.0000 1.0000, fval = 8.1777e-10, 85 iterations, 159 function evaluations."""
    x, fv, flag, it, ne = o.fminsearch(rosenbrock, [-1.2, 1.0])
    assert '%.4e' % fv == '8.1777e-10' and flag == 1 and it == 85 and ne == 159
    assert np.allclose(x, [1.0, 1.0], atol=5e-5)


class _Analytic:
    """A context whose acquisition is an analytic function of the candidate (maximised by fminsearch as -f)."""

    def __init__(self, f):
        self.f = f
        self.calls = 0

    def set_candidates(self, P):
        self.P = np.array(P, dtype=np.float64)

    def posterior(self):
        self.calls += 1

    def acquire(self, acq, kappa, mode, inc):
        pass

    def read(self, mu=True, var=True, acq=False):
        return np.array([-self.f(p) for p in self.P])


def test_batched_search_walks_the_sequential_paths():
    """Every start of the lockstep version ends at the same point, value, iteration and evaluation counts as the sequential
    restatement run on it alone -- including starts with zero coordinates, starts that shrink, and starts that stop early or
    hit the limits while others go on."""
    funcs = [rosenbrock,
             lambda x: (x[0] - 0.3) ** 2 + 3 * (x[1] + 0.2) ** 2 + 0.5 * np.sin(5 * x[0]) * np.cos(3 * x[1]),
             lambda x: np.abs(x[0]) + np.abs(x[1] - 1.0) + 0.1 * np.abs(x[0] * x[1])]
    rng = np.random.default_rng(0)
    X0 = np.vstack([[-1.2, 1.0], [0.0, 0.0], [0.0, 2.0], rng.uniform(-2, 2, (9, 2))])
    for f in funcs:
        for kw in (dict(), dict(tol_x=1e-8, tol_fun=1e-8), dict(max_iter=30), dict(max_fun_evals=25)):
            g = _Analytic(f)
            x, val, flag, it, ne = acqopt.fminsearch(g, X0, acq=1, **kw)
            for r in range(len(X0)):
                xs, fs, fl, its, nes = o.fminsearch(f, X0[r], **kw)
                assert np.array_equal(x[r], xs) and val[r] == -fs and flag[r] == fl and it[r] == its and ne[r] == nes, (r, kw)
            assert g.calls <= 3 * int(it.max()) + 1       # at most three evaluation rounds per lockstep iteration


def test_one_dimensional_and_three_dimensional_starts():
    f1 = lambda x: (x[0] - 2.0) ** 2
    f3 = lambda x: np.sum((x - np.array([0.5, -0.25, 1.5])) ** 2) + 0.1 * np.sin(4 * x[0])
    for f, X0 in ((f1, np.array([[0.0], [5.0], [-3.0]])), (f3, np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]))):
        x, val, flag, it, ne = acqopt.fminsearch(_Analytic(f), X0, acq=1, tol_x=1e-8, tol_fun=1e-8)
        for r in range(len(X0)):
            xs, fs, fl, its, nes = o.fminsearch(f, X0[r], tol_x=1e-8, tol_fun=1e-8)
            assert np.array_equal(x[r], xs) and val[r] == -fs and (flag[r], it[r], ne[r]) == (fl, its, nes)
