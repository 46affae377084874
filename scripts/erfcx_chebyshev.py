#!/usr/bin/env python3
"""Coefficients of erfcx_over_t (bayesian_optimization_amd/csrc/k_acquisition.hpp): the Chebyshev series of
    G(u) = exp(z^2) erfc(z) / t,   t = 2 / (2 + z),   u = 2 t - 1 in (-1, 1]   (z = 0 at u = 1, z = inf at u = -1)
projected at 96 Chebyshev nodes in 60-digit arithmetic, truncated to 28 terms and re-expanded in powers of u (the same
polynomial; the kernel evaluates it by Horner's rule, one FMA per term); prints the truncation error over a grid, the
power-basis table as it stands in the source, and the rounding error of Horner's rule with that table in binary64
(exact rational emulation of the FMA) beside that of Clenshaw's recurrence on the Chebyshev form.
    python3 scripts/erfcx_chebyshev.py   (build container: needs mpmath)"""
from fractions import Fraction as F

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
# the truncated series in powers of u: T_k by its three-term recurrence, all in 60-digit arithmetic
T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
for k in range(2, M):
    nxt = [mp.mpf(0)] + [2 * x for x in T[-1]]
    for i, x in enumerate(T[-2]):
        nxt[i] -= x
    T.append(nxt)
a = [mp.mpf(0)] * M
for k in range(M):
    for j, x in enumerate(T[k]):
        a[j] += cc[k] * x
print('sum of |a_j| = %.4f (G(1) = 1: no cancellation to speak of)' % float(mp.fsum(abs(x) for x in a)))
print(',\n'.join('    %s' % mp.nstr(x, 20) for x in a))
ad, cd = [float(x) for x in a], [float(x) for x in cc]


def horner64(u):
    p = ad[M - 1]
    for j in range(M - 2, -1, -1):
        p = float(F(p) * F(u) + F(ad[j]))           # one correctly rounded FMA
    return p


def clenshaw64(u):
    b1 = b2 = 0.0
    for k in range(M - 1, 0, -1):
        b1, b2 = float(F(2.0 * u) * F(b1) + F(float(F(cd[k]) - F(b2)))), b1
    return float(F(u) * F(b1) + F(float(F(cd[0]) - F(b2))))


wh = wc = 0.0
for i in range(2001):
    u = -1.0 + i / 1000.0
    g = G(mp.mpf(u))
    wh = max(wh, float(abs(mp.mpf(horner64(u)) - g) / g))
    wc = max(wc, float(abs(mp.mpf(clenshaw64(u)) - g) / g))
print('binary64 evaluation against G itself, 2001 points: Horner (power basis) %.3e, Clenshaw (Chebyshev basis) %.3e' % (wh, wc))
