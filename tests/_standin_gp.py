"""Stand-in for a GPU context in launch rehearsals (bench.py --rehearse, CPU tests): the posterior and the acquisition of
candidate j are fixed functions of the GLOBAL index j, so the sharded result can be checked against a single pass."""
import numpy as np


def mu_of(j):
    return np.cos(0.11 * np.asarray(j, dtype=np.float64))


def acq_of(j, incumbent):
    j = np.asarray(j, dtype=np.float64)
    a = np.sin(0.37 * j) - 0.01 * incumbent
    return np.where(np.isin(j, (100, 900)), 5.0, a)       # an exact tie between two blocks: the lower index must win


class StandinGP:
    device = None

    def __init__(self, lo, hi):
        self.j = np.arange(lo, hi)

    def posterior(self):
        return float(mu_of(self.j).max())

    def acquire(self, acq, kappa, incumbent_mode, incumbent):
        a = acq_of(self.j, incumbent)
        i = int(np.argmax(a))
        return i, float(a[i])
