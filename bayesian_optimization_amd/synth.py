"""Deterministic synthetic inputs for the benchmark configurations.

Counter-based splitmix64: element i of stream `seed` depends only on (seed, i), so every rank can
generate its own slice of a candidate set without communication and independent of NumPy's RNG."""
from __future__ import annotations

import numpy as np


def uniform(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """count doubles in [0,1): (splitmix64(seed, offset+i) >> 11) * 2^-53."""
    i = np.arange(offset, offset + count, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, count: int) -> np.ndarray:
    """Box-Muller on two uniform streams."""
    u1 = uniform(seed, count)
    u2 = uniform(seed + 7919, count)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


CONFIGS = {
    # "Synthetic GP, n=4096 train, d=4, RBF, 1M-candidate predict+EI on one MI355X"
    'cfg2': dict(n=4096, d=4, m=1 << 20, kernel=0, theta=0.35, sigmaY=0.1, kappa=0.01, seed_x=1, seed_c=2),
    # "n=32768 train, d=8, Matern-5/2, 64M candidates sharded across 1/2/4/8 GPUs"
    'cfg3': dict(n=32768, d=8, m=1 << 26, kernel=1, theta=0.6, sigmaY=0.1, kappa=0.01, seed_x=3, seed_c=33),
}


def training_set(cfg):
    n, d = cfg['n'], cfg['d']
    X = uniform(cfg['seed_x'], n * d).reshape(n, d)
    y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * normal(cfg['seed_x'] + 100, n)
    return X, y, np.full(d, cfg['theta'])


def candidates(cfg, lo: int, hi: int) -> np.ndarray:
    """Rows [lo, hi) of the configuration's candidate set."""
    d = cfg['d']
    return uniform(cfg['seed_c'], (hi - lo) * d, offset=lo * d).reshape(hi - lo, d)
