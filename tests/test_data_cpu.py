"""Host-side data plumbing of the measured-mixture experiment (no GPU): objective from spectra, candidate grid, snapping."""
import numpy as np
import pytest

from bayesian_optimization_amd import data
from conftest import golden as load_golden


def test_candidate_grid_matches_the_reference_fixture():
    g = load_golden('g1_mat_mixture')
    sx, sy, xp = data.candidate_grid(g['x_all'], g['xTrain'])
    assert np.array_equal(sx, g['grid_sx']) and np.array_equal(sy, g['grid_sy'])
    assert np.array_equal(xp, g['xTest'])                       # 76 x 80 = 6080 candidates, first axis slowest


def test_mixture_objective_formula_and_shapes():
    rng = np.random.default_rng(0)
    IM = rng.normal(size=(9, 30))
    Cc = rng.uniform(size=(2, 9))
    x, y = data.mixture_objective(IM, Cc, p=4)
    assert x.shape == (9, 2) and y.shape == (9, 1)
    for i in range(9):
        assert y[i, 0] == -np.mean((IM[i] - IM[4]) ** 2)
        assert np.array_equal(x[i], Cc[:, i] * 1e5)
    assert y[4, 0] == 0.0 and np.argmax(y) == 4
    with pytest.raises(ValueError):
        data.mixture_objective(IM, Cc.T, p=4)


def test_split_follows_the_permutation():
    g = load_golden('g1_mat_mixture')
    xT, yT, xTest, yTest = data.train_test_split(g['x_all'], g['y_all'], 40, g['perm'], nT=94)
    assert np.array_equal(xT, g['xTrain']) and np.array_equal(yT, g['yTrain'])


def test_pool_snaps_to_the_nearest_measured_point_and_shrinks():
    xdat = np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 0.5], [1.0, 1.0]])
    ydat = np.array([10.0, 11.0, 12.0, 13.0])
    pool = data.MeasuredPool(xdat, ydat)
    xs, ys = pool.take_nearest([0.9, 1.2])
    assert np.array_equal(xs, [1.0, 1.0]) and ys[0] == 11.0 and len(pool) == 3      # first of the two equal points
    xs, ys = pool.take_nearest([0.9, 1.2])
    assert ys[0] == 13.0 and len(pool) == 2
    xs, ys = pool.take_nearest([5.0, 5.0])
    assert ys[0] == 12.0
    pool.take_nearest([0, 0])
    with pytest.raises(IndexError):
        pool.take_nearest([0, 0])
