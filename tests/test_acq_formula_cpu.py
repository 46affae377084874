"""CPU: the coefficient table of the acquisition kernels' complementary error function (erfcx_over_t in csrc/k_acquisition.hpp),
read from the source and evaluated with the kernel's own arithmetic in NumPy, against scipy's ndtr (what the reference's
norm.cdf calls, BayesOpt_v4.ipynb:194) and against the reference's EI formula."""
import os
import re

import numpy as np
from scipy.stats import norm

from conftest import ROOT
from oracle import gp_oracle as o


def table():
    src = open(os.path.join(ROOT, 'bayesian_optimization_amd', 'csrc', 'k_acquisition.hpp')).read()
    blk = src[src.index('__constant__ double ERFCX_POW[28] = {'):]
    c = np.array([float(x) for x in re.sub(r'\s+', ' ', blk[blk.index('{') + 1:blk.index('}')]).split(',')])
    assert c.shape == (28,)
    return c


def phi_parts(Z, c):
    """(Phi, density exponential) with the device's operations (Horner's rule on the power-basis table, one exponential, the
    Z * Z rounding put back); NumPy has no fused multiply-add, so every Horner step rounds twice where the device rounds once."""
    z = np.abs(Z) * 0.70710678118654752440
    t = 2.0 / (2.0 + z)
    u = 2.0 * t - 1.0
    g = np.full_like(u, c[27])
    for k in range(26, -1, -1):
        g = g * u + c[k]
    ZZ = Z * Z
    E = np.exp(-ZZ / 2.0)
    with np.errstate(invalid='ignore'):
        e = np.where(np.isfinite(ZZ), (Z.astype(np.longdouble) * Z.astype(np.longdouble) - ZZ.astype(np.longdouble)).astype(np.float64), 0.0)
    Ec = E - E * (0.5 * e)             # the rounding of Z * Z put back (one FMA on the device)
    h = 0.5 * t * Ec * g
    return np.where(Z < 0, h, 1.0 - h), E


def test_normal_cdf_formula_against_scipy():
    c = table()
    Z = np.concatenate([np.linspace(-37, 8.2, 200001), -np.logspace(-14, 1.5, 3000), np.logspace(-14, 0.9, 3000)])
    got, _ = phi_parts(Z, c)
    ref = norm.cdf(Z)
    rel = np.abs(got - ref) / ref
    assert np.max(rel[np.abs(Z) <= 2]) <= 3e-15 and np.max(rel[np.abs(Z) <= 6]) <= 1e-14
    # further out scipy's own exponent, exp(-a * a) with a = Z / sqrt(2) rounded, is Z^2 ulp from the exact one: that is the
    # whole difference (checked against 40-digit values below: the formula here is the closer one)
    assert np.max(rel) <= 4e-13
    import mpmath as mp
    mp.mp.dps = 40
    for zz in (-3.0, -8.0, -15.7165691034674, -20.0, -30.0, -36.5):
        exact = mp.ncdf(mp.mpf(zz))
        mine = abs(mp.mpf(float(phi_parts(np.array([zz]), c)[0][0])) - exact) / exact
        theirs = abs(mp.mpf(float(norm.cdf(zz))) - exact) / exact
        assert mine <= 1e-15 and mine <= theirs
    assert phi_parts(np.array([-np.inf, np.inf, 0.0]), c)[0].tolist() == [0.0, 1.0, 0.5]
    assert np.isnan(phi_parts(np.array([np.nan]), c)[0][0])


def test_ei_formula_against_the_reference_expression():
    c = table()
    rng = np.random.default_rng(2)
    mu = rng.standard_normal(20000)
    var = rng.random(20000) ** 3
    for inc, kappa in ((float(mu.max()), 0.01), (0.3, 0.0), (-1.0, 0.5)):
        sigma = np.sqrt(var)
        imp = mu - inc - kappa
        Phi, E = phi_parts(imp / sigma, c)
        got = imp * Phi + sigma * (E * 0.39894228040143267794)
        ref = o.expected_improvement(mu.copy(), var, kappa, incumbent=inc)
        assert np.max(np.abs(got - ref)) <= 1e-15 * np.max(np.abs(ref))
