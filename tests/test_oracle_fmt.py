"""CPU restatement of the reference's planner helpers (halton, sample_free, FMT*): known values and path properties.
Parity vs the reference is unpinned -- it records no FMT* output and its example cannot terminate."""
import numpy as np

EXAMPLE = [[(-0.5, 0), (1, 0), (1, 1), (0, 1)], [(0, -0.5), (1, 0), (1, 1), (0, 1)], [(-0.6, 0.148), (-1, 0.148), (-1, 0), (-0.6, 0)]]


def _world(polys):
    lines, off = [], [0]
    for q in polys:
        for i in range(len(q)):
            lines.append(tuple(q[i]) + tuple(q[(i + 1) % len(q)]))
        off.append(len(lines))
    return np.array(lines, np.float32).reshape(-1, 4), np.array(off, np.int32)


def _radical_inverse(k, b):
    f, r = 1.0, 0.0
    while k > 0:
        f /= b
        r += f * (k % b)
        k //= b
    return r


def test_halton_is_the_radical_inverse_and_resumes(oracle):
    for b in (2, 3, 5):
        a, st = oracle.halton(b, 50)
        assert np.allclose(a, [_radical_inverse(k, b) for k in range(1, 51)], atol=1e-7)
        more, _ = oracle.halton(b, 10, st)                      # continues from the saved state (:106-109)
        assert np.allclose(more, [_radical_inverse(k, b) for k in range(51, 61)], atol=1e-7)


def test_sample_free_and_fmt_star_example_world(oracle):
    lines, off = _world(EXAMPLE)
    pts, hs = oracle.sample_free(200, (-1, 1, -1, 1), lines, off)
    assert pts.shape == (200, 2) and np.all(pts[0] == 0) and np.all(np.abs(pts) <= 1)
    assert len({tuple(p) for p in pts}) == 200                  # Halton points do not repeat
    assert hs[1] >= 256                                          # more than 199 candidates were drawn: some were rejected
    r = oracle.fmt_star(pts, (-0.5, 1.0), (1.0, -1.0), 1.0, lines)      # the query of examples/test.cpp:284
    assert r["status"] == 0 and r["len"] >= 3
    p = r["path"].astype(np.float64)
    assert np.allclose(p[0], (-0.5, 1)) and np.allclose(p[-1], (1, -1))
    legs = np.hypot(*np.diff(p, axis=0).T)
    assert abs(legs.sum() - r["cost"]) < 1e-5 and legs.max() <= 1.0 + 1e-6       # radius rn^2 = 1
    straight = np.hypot(1.5, 2.0)
    assert straight <= r["cost"] < 1.5 * straight
    # every leg misses every obstacle edge (orientation test independent of the oracle's own intersects)
    def crosses(a, b, c, d):
        o = lambda p, q, r_: np.sign((q[0] - p[0]) * (r_[1] - p[1]) - (q[1] - p[1]) * (r_[0] - p[0]))
        return o(a, b, c) * o(a, b, d) < 0 and o(c, d, a) * o(c, d, b) < 0
    for i in range(len(p) - 1):
        assert not any(crosses(p[i], p[i + 1], l[:2], l[2:]) for l in lines)
    # smaller radius: longer chain, never cheaper
    r2 = oracle.fmt_star(pts, (-0.5, 1.0), (1.0, -1.0), 0.6, lines)
    assert r2["status"] == 0 and r2["len"] > r["len"] and r2["cost"] >= r["cost"] - 1e-6
    # goal walled in: no path
    box, boff = _world([[(0.5, 0.5), (0.9, 0.5), (0.9, 0.9), (0.5, 0.9)]])
    pts2, _ = oracle.sample_free(100, (-1, 1, -1, 1), box, boff)
    assert oracle.fmt_star(pts2, (-0.5, -0.5), (0.7, 0.7), 1.0, box)["status"] == 1
