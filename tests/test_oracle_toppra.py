"""Pin the TOPP-RA oracle against the reference's only recorded output
(examples/output.json -> tests/golden/toppra_1dof_output.npz)."""
import os

import numpy as np


def _nudge(pos, arclength):
    """bezier_spline::resample(..., nudge_positions=true) mutates prof.pos[0] in place
    before it is serialised (sea_current.hpp:902-912; zmq_test.cpp:91).  Restated here
    (float32, sequential) so the recorded `pos` can be compared too."""
    p = pos.astype(np.float32).copy()
    p[0] = 0
    p[-1] = arclength
    for i in range(1, len(p) - 1):
        if p[i] < p[i - 1] or p[i] > p[i + 1]:
            p[i] = (p[i - 1] + p[i + 1]) / np.float32(2)
        if p[i] < 0:
            p[i] = 0
        if p[i] > arclength:
            p[i] = arclength
    return p


def test_toppra_1dof_fixture(oracle, golden_dir):
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    L = float(fx["arclength"])  # pos_end = ad.arclength (float -> value_type), zmq_test.cpp:69
    r = oracle.toppra([0.0], [L], [0.0], [0.0], [fx["vel_lim"][0]], [fx["vel_lim"][1]],
                      [fx["acc_lim"][0]], [fx["acc_lim"][1]], N=100)
    assert r["status"] == 0
    T = r["t"][-1]
    # duration: fixture stores float32(T)
    assert np.float32(T) == fx["time"][-1]
    dt = float(np.float32(0.02))  # `const float dt=0.02` promoted to double, sea_current.hpp:1199,1237
    s = oracle.toppra_sample([0.0], [L], [0.0], [0.0], r["x"], r["t"], dt)
    assert s["length"] == fx["time"].shape[0] == 4328
    assert np.array_equal(s["time"].astype(np.float32), fx["time"])
    rel = lambda a, b: np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b))
    # north_star tolerance: 1e-5 relative on velocity profiles; we are at float32 round-off
    assert rel(s["vel"][0], fx["vel"]) < 2e-7
    assert rel(s["acc"][0], fx["acc"]) < 2e-7
    assert rel(_nudge(s["pos"][0], fx["arclength"]), fx["pos"]) < 2e-7
    # the recorded extrema quoted in BASELINE.md
    assert abs(s["vel"][0].max() - 0.253140) < 1e-6
    assert abs(np.abs(s["acc"][0]).max() - 0.497336) < 1e-6


def test_toppra_constraints_hold_6dof(oracle):
    rng = np.random.default_rng(7)
    dof, N = 6, 200
    vl = np.array([2, 2, 2, 3, 3, 3.0]); al = np.array([5, 5, 5, 8, 8, 8.0])
    for _ in range(20):
        p0 = rng.uniform(-np.pi, np.pi, dof); p1 = rng.uniform(-np.pi, np.pi, dof)
        v0 = rng.uniform(-1, 1, dof); v1 = rng.uniform(-1, 1, dof)
        r = oracle.toppra(p0, p1, v0, v1, -vl, vl, -al, al, N=N)
        assert r["status"] == 0
        x, u, K = r["x"], r["u"], r["K"]
        assert np.all(x >= -1e-12) and np.all(x <= K[:, 1] + 1e-9) and np.all(x >= K[:, 0] - 1e-9)
        s = np.arange(N + 1) / N
        d = p1 - p0
        c2 = 3 * d - 2 * v0 - v1; c3 = -2 * d + v0 + v1
        qs = v0[None] + s[:, None] * (2 * c2[None] + 3 * c3[None] * s[:, None])
        qss = 2 * c2[None] + 6 * c3[None] * s[:, None]
        # velocity limits at every gridpoint
        assert np.all(np.abs(qs) * np.sqrt(np.maximum(x, 0))[:, None] <= vl[None] * (1 + 1e-9) + 1e-9)
        # acceleration limits at collocation points i < N
        a = qs[:-1] * u[:, None] + qss[:-1] * x[:-1, None]
        assert np.all(np.abs(a) <= al[None] * (1 + 1e-9) + 1e-9)
        # time-optimality signature: forward pass saturates some constraint or the K bound
        assert np.all(np.diff(r["t"]) > 0)
