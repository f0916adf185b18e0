"""Pin the Bezier smoothing / arclength oracle against the reference's recorded
output (examples/output.json -> tests/golden/toppra_1dof_output.npz: `arclength.*`, pos_x/pos_y)."""
import os

import numpy as np


def test_from_path_and_arclength_fixture(oracle, golden_dir):
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    ctrl = oracle.bezier_from_path(fx["waypoints"])  # (0,0),(10,0),(10,10): examples/zmq_test.py:7-10
    # control points by hand from sea_current.hpp:343-377: |T| = 5 everywhere, T1 at 45 degrees
    r = 5 / np.sqrt(2)
    expect = np.array([[[0, 0], [5, 0], [10 - r, -r], [10, 0]], [[10, 0], [10 + r, r], [10, 5], [10, 10]]], np.float32)
    assert np.allclose(ctrl, expect, atol=1e-6)
    total, cum = oracle.bezier_arclength(ctrl, 100)  # precision 0.01 (:765)
    assert abs(total - float(fx["arclength"])) / total < 2e-7            # fixture is float32-accumulated
    assert np.abs(cum - fx["arclength_segments"]).max() < 5e-6
    assert np.allclose(fx["arclength_positions"], np.minimum(np.arange(101) * np.float32(0.01), 1)[None])
    # curve extrema seen in the recorded resampled positions
    t = np.linspace(0, 1, 4001)
    p0 = oracle.bezier_eval(ctrl, np.zeros(4001, np.int32), t)
    p1 = oracle.bezier_eval(ctrl, np.ones(4001, np.int32), t)
    assert abs(p0[:, 1].min() - fx["pos_y"].min()) < 2e-6 and abs(p1[:, 0].max() - fx["pos_x"].max()) < 2e-6


def test_hodograph_is_derivative(oracle):
    rng = np.random.default_rng(3)
    ctrl = rng.uniform(-2, 2, (5, 4, 2)).astype(np.float32)
    seg = rng.integers(0, 5, 200).astype(np.int32)
    t = rng.uniform(0.01, 0.99, 200)
    h = 1e-6
    d1 = oracle.bezier_eval(ctrl, seg, t, 1)
    fd = (oracle.bezier_eval(ctrl, seg, t + h) - oracle.bezier_eval(ctrl, seg, t - h)) / (2 * h)
    assert np.allclose(d1, fd, rtol=1e-6, atol=1e-6)
    d2 = oracle.bezier_eval(ctrl, seg, t, 2)
    fd2 = (oracle.bezier_eval(ctrl, seg, t + h, 1) - oracle.bezier_eval(ctrl, seg, t - h, 1)) / (2 * h)
    assert np.allclose(d2, fd2, rtol=1e-5, atol=1e-5)


def test_shrink_tangent_against_obstacle_edge(oracle):
    # a wall just right of the first waypoint cuts its tangent (sea_current.hpp:575-596)
    path = np.array([[0, 0], [10, 0], [10, 10]], np.float32)
    free = oracle.bezier_from_path(path)
    wall = np.array([[2, -1, 2, 1]], np.float32)
    cut = oracle.bezier_from_path(path, lines=wall)
    assert np.allclose(free[0, 1], [5, 0]) and np.allclose(cut[0, 1], [2, 0], atol=1e-6)
    assert np.allclose(cut[1], free[1])  # other tangents untouched


def test_straight_path_arclength_is_length(oracle):
    path = np.array([[0, 0], [3, 4]], np.float32)
    total, cum = oracle.bezier_arclength(oracle.bezier_from_path(path), 100)
    assert abs(total - 5.0) < 1e-9 and abs(cum[0, -1] - 5.0) < 1e-9
