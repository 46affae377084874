#!/usr/bin/env python3
"""Coefficients of erfcx_over_t (bayesian_optimization_amd/csrc/k_acquisition.hpp): the Chebyshev series of
    G(u) = exp(z^2) erfc(z) / t,   t = 2 / (2 + z),   u = 2 t - 1 in (-1, 1]   (z = 0 at u = 1, z = inf at u = -1)
projected at 96 Chebyshev nodes in 60-digit arithmetic, truncated to 28 terms; prints the truncation error over a grid
and the table as it stands in the source.      python3 scripts/erfcx_chebyshev.py   (build container: needs mpmath)"""
import mpmath as mp

mp.mp.dps = 60


def G(u):
    t = (u + 1) / 2
    if t == 0:
        return 1 / (2 * mp.sqrt(mp.pi))
    z = 2 * (1 - t) / t
    if z > 200:       # asymptotic series of erfcx: far more accurate there than exp(z^2) erfc(z) needs to be
        s, term = mp.mpf(1), mp.mpf(1)
        for k in range(1, 40):
            term *= -(2 * k - 1) / (2 * z * z)
            s += term
        return s / (z * mp.sqrt(mp.pi)) / t
    return mp.exp(z * z) * mp.erfc(z) / t


N, M = 96, 28
nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / N) for k in range(N)]
vals = [G(x) for x in nodes]
c = [2 * mp.fsum(vals[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / N) for k in range(N)) / N for j in range(N)]
c[0] /= 2
cc = c[:M]


def clenshaw(u):
    b1 = b2 = mp.mpf(0)
    for k in range(M - 1, 0, -1):
        b1, b2 = 2 * u * b1 - b2 + cc[k], b1
    return u * b1 - b2 + cc[0]


worst = max(abs(clenshaw(u) - G(u)) / G(u) for u in (mp.mpf(-1) + mp.mpf(2) * i / 4000 for i in range(4001)))
print('max relative truncation error with %d terms: %.3e' % (M, float(worst)))
print(',\n'.join('    %s' % mp.nstr(x, 20) for x in cc))
