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


def test_resample_fixture(oracle, golden_dir):
    """resample(prof.pos[0], ad, true) + angular_velocity of the recorded run (examples/zmq_test.cpp:91-93): the JSON's
    pos_x / pos_y and ang_vel are reproduced from its `pos` and `vel` series."""
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    ctrl = oracle.bezier_from_path(fx["waypoints"])
    r = oracle.bezier_resample(ctrl, fx["arclength_segments"], fx["arclength"], fx["pos"], nudge=True)
    assert r["status"] == 0
    assert np.array_equal(r["pos"], fx["pos"])                  # the recorded series is already the nudged one
    assert np.abs(r["pts"][:, 0] - fx["pos_x"]).max() < 2e-5    # coordinates up to 11: 1e-6 relative (float32 fit upstream)
    assert np.abs(r["pts"][:, 1] - fx["pos_y"]).max() < 2e-5
    assert np.abs(fx["vel"] * r["curvature"] - fx["ang_vel"]).max() < 3e-7
    assert np.bincount(r["seg"]).tolist() == [2164, 2164]       # boundary sample duplicated, last sample dropped (:938-993)
    # same through the oracle's own arclength tables
    tot, cum = oracle.bezier_arclength(ctrl, 100)
    r2 = oracle.bezier_resample(ctrl, cum.astype(np.float32), np.float32(tot), fx["pos"], nudge=True)
    assert np.abs(r2["pts"] - np.stack([fx["pos_x"], fx["pos_y"]], 1)).max() < 2e-5


def test_resample_nudge_and_split(oracle):
    """nudge semantics (:902-913) and the overlapping split on a 4-segment spline, against a literal numpy replay."""
    rng = np.random.default_rng(12)
    path = np.array([[0, 0], [3, 1], [5, 4], [9, 3], [12, 6]], np.float32)
    ctrl = oracle.bezier_from_path(path)
    tot, cum = oracle.bezier_arclength(ctrl, 40)
    cum = cum.astype(np.float32)
    AL = np.float32(cum[:, -1].sum())
    n = 700
    pp = (np.linspace(0, 1, n) ** 1.3 * AL).astype(np.float32)
    bad = rng.choice(np.arange(2, n - 2), 25, replace=False)
    pp[bad] += rng.normal(0, 0.4, 25).astype(np.float32)       # non-monotone glitches, some beyond [0, AL]
    pp[5] = -1.0
    pp[n - 4] = AL + 1
    ref = pp.copy()
    ref[0] = 0; ref[-1] = AL
    for i in range(1, n - 1):
        if ref[i] < ref[i - 1] or ref[i] > ref[i + 1]:
            ref[i] = (ref[i - 1] + ref[i + 1]) / np.float32(2)
        ref[i] = min(max(ref[i], np.float32(0)), AL)
    r = oracle.bezier_resample(ctrl, cum, AL, pp, nudge=True)
    assert r["status"] == 0 and np.array_equal(r["pos"], ref)
    assert np.all(np.diff(r["seg"]) >= 0) and r["seg"][0] == 0 and r["seg"][-1] == 3
    assert np.all((r["t"] >= 0) & (r["t"] <= 1))
    # points lie on the curve at the reported parameter, and their arclength position tracks the profile
    p = oracle.bezier_eval(ctrl, r["seg"], r["t"].astype(np.float64))
    assert np.abs(p - r["pts"]).max() < 1e-6
    lens = cum[:, -1].astype(np.float64)
    tk = np.arange(41) / 40
    s_of = np.array([lens[:g].sum() + np.interp(t, tk, cum[g]) for g, t in zip(r["seg"], r["t"])])
    want = np.concatenate([ref[:1], ref]).astype(np.float64)[np.arange(n) - r["seg"] + 1]   # output o of segment i is sample o - i
    assert np.abs(s_of - want).max() < 0.02 * AL


def test_chebfit_is_the_least_squares_solution(oracle):
    """The free chebfit (sea_current.hpp:1109-1138): `degree` Chebyshev columns of the abscissa normalised by its own
    range; the restatement's Householder solution equals numpy's least-squares solution of the same system."""
    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(-2, 7, 500)).astype(np.float32)
    y = (np.sin(x) + 0.1 * x * x).astype(np.float32)
    for degree in (1, 2, 5, 10):
        coef, xmin, xmax = oracle.chebfit(x, y, degree)
        assert xmin == x.min() and xmax == x.max()
        xn = (2 * x.astype(np.float64) - (xmax + xmin)) / (xmax - xmin)
        T = np.polynomial.chebyshev.chebvander(xn, degree - 1)
        ref = np.linalg.lstsq(T, y.astype(np.float64), rcond=None)[0]
        assert np.allclose(coef, ref, rtol=1e-9, atol=1e-9)
        assert np.allclose(oracle.chebeval(x, coef, xmin, xmax), T @ ref, rtol=1e-9, atol=1e-9)


def test_general_degree_curve_matches_bernstein_sum(oracle):
    """bezier_curve for any control polygon (:700-763): de Casteljau equals the Bernstein sum; degree 3 equals the cubic
    evaluation that is pinned to the recording; the hodograph's control points give the derivative."""
    from math import comb
    rng = np.random.default_rng(6)
    t = np.linspace(0, 1, 41)
    for deg in (1, 2, 3, 7):
        ctrl = rng.uniform(-3, 3, (2, deg + 1, 2)).astype(np.float32)
        seg = (np.arange(41) % 2).astype(np.int32)
        got = oracle.bezier_curve(ctrl, seg, t)
        B = np.stack([comb(deg, k) * t**k * (1 - t) ** (deg - k) for k in range(deg + 1)], 1)
        want = np.einsum("mk,mkc->mc", B, ctrl[seg].astype(np.float64))
        assert np.allclose(got, want, atol=1e-12)
    ctrl = rng.uniform(-3, 3, (3, 4, 2)).astype(np.float32)
    seg = (np.arange(41) % 3).astype(np.int32)
    assert np.allclose(oracle.bezier_curve(ctrl, seg, t), oracle.bezier_eval(ctrl, seg, t, 0), atol=1e-12)
    hod = (3 * (ctrl[:, 1:] - ctrl[:, :-1])).astype(np.float32)          # degree * (P[j+1] - P[j]), :1046
    assert np.allclose(oracle.bezier_curve(hod, seg, t), oracle.bezier_eval(ctrl, seg, t, 1), atol=1e-5)


def test_shrink_tangent_on_its_own_closed_forms(oracle):
    """sco_bezier_shrink_tangent (sea_current.hpp:575-596) against geometry done by hand: a wall in front of the tangent cuts it at
    the wall, a wall behind the waypoint cuts it by the mirrored stretch, a far wall leaves k * T, two walls apply in order."""
    T = np.array([[4, 0], [4, 0], [4, 0], [0, 3], [4, 0]], np.float32)
    Wp = np.array([[0, 0], [0, 0], [0, 0], [1, 1], [0, 0]], np.float32)
    wall = lambda x, y0, y1: [x, y0, x, y1]
    # k = 0.5: the stretch W + 2 ex reaches x = 2
    assert np.allclose(oracle.bezier_shrink_tangent(T[:1], Wp[:1], 0.5, [wall(1.5, -1, 1)]), [[1.5, 0]])           # cut at the wall
    assert np.allclose(oracle.bezier_shrink_tangent(T[1:2], Wp[1:2], 0.5, [wall(-1.25, -1, 1)]), [[1.25, 0]])      # wall behind: W - T crosses it
    assert np.allclose(oracle.bezier_shrink_tangent(T[2:3], Wp[2:3], 0.5, [wall(2.5, -1, 1)]), [[2.0, 0]])         # out of reach: k T
    assert np.allclose(oracle.bezier_shrink_tangent(T[3:4], Wp[3:4], 1.0, [[0, 2.5, 2, 2.5]]), [[0, 1.5]])          # vertical tangent, horizontal wall
    assert np.allclose(oracle.bezier_shrink_tangent(T[4:5], Wp[4:5], 0.5, [wall(1.5, -1, 1), wall(-1.0, -1, 1)]), [[1.0, 0]])   # first the front wall, then the rear one cuts further
    assert np.allclose(oracle.bezier_shrink_tangent(T[:1], Wp[:1], 0.5, [[1.5, 0.5, 1.5, 2.0]]), [[2.0, 0]])        # the wall does not reach the tangent's line


def test_smooth_one_reproduces_the_recorded_run(oracle, golden_dir):
    """oracle.smooth_one (the CPU composition the batched GPU sequence is checked against) on the reference's own request
    (examples/zmq_test.py:7-10) against examples/output.json."""
    import os
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    r = oracle.smooth_one(fx["waypoints"], vmax=float(fx["vel_lim"][1]), amax=float(fx["acc_lim"][1]), dt=0.02, N=100)
    assert r["status"] == 0 and r["toppra_status"] == 0 and r["length"] == fx["pos"].shape[0]
    assert abs(float(r["arclength"]) - float(fx["arclength"])) < 2e-6 * float(fx["arclength"])
    assert np.abs(r["pts"][:, 0] - fx["pos_x"]).max() < 5e-5 and np.abs(r["pts"][:, 1] - fx["pos_y"]).max() < 5e-5
    assert np.abs(r["vel"] * r["curvature"] - fx["ang_vel"]).max() < 2e-6
