"""GPU parity: the reference's own planner (FMT* over Halton samples) batched on the GPU vs its CPU
restatement -- node-for-node identical paths and bit-equal float costs.  Parity vs the reference itself is unpinned (it
holds no recorded FMT* output); the world of examples/test.cpp:249-259 is used as input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _world(polys):
    lines, off = [], [0]
    for q in polys:
        for i in range(len(q)):
            lines.append(tuple(q[i]) + tuple(q[(i + 1) % len(q)]))
        off.append(len(lines))
    return np.array(lines, np.float32).reshape(-1, 4), np.array(off, np.int32)


EXAMPLE = [[(-0.5, 0), (1, 0), (1, 1), (0, 1)], [(0, -0.5), (1, 0), (1, 1), (0, 1)], [(-0.6, 0.148), (-1, 0.148), (-1, 0), (-0.6, 0)]]


@pytest.mark.parametrize("n,rn", [(200, 1.0), (200, 0.6), (1000, 0.45), (30, 0.7), (900, 1.0)])   # (900, 1.0): more samples in range than a neighbour list holds (the scans instead)
def test_fmt_star_matches_oracle(oracle, n, rn):
    import torch
    import sea_current_amd as sc
    ctx = sc.Context(0)
    lines, off = _world(EXAMPLE)
    samples, hs = oracle.sample_free(n, (-1, 1, -1, 1), lines, off)
    assert samples.shape == (n, 2) and hs[1] > 0
    rng = np.random.default_rng(n)
    starts = [(-0.5, 1.0)] + [tuple(samples[i]) for i in rng.integers(1, n, 12)] + [(-0.9, -0.9), (0.5, 0.5)]
    goals = [(1.0, -1.0)] + [tuple(samples[i]) for i in rng.integers(1, n, 12)] + [(0.5, 0.5), (-0.9, 0.9)]   # two with an endpoint inside an obstacle
    Q = len(starts)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32)).cuda()
    out = ctx.fmt_star(t(samples), t(starts), t(goals), rn, t(lines), Lmax=128)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    nok = 0
    for q in range(Q):
        ref = oracle.fmt_star(samples, starts[q], goals[q], rn, lines, Lmax=128)
        assert got["status"][q] == ref["status"], q
        if ref["status"] == 0:
            nok += 1
            assert got["len"][q] == ref["len"] and got["cost"][q] == np.float32(ref["cost"]), q
            assert np.array_equal(got["path"][q, :ref["len"]], ref["path"]), q
            pth = ref["path"]                                     # sanity of the restatement itself: legs are collision free
            assert np.allclose(pth[0], starts[q]) and np.allclose(pth[-1], goals[q])
    assert nok >= (Q - 2) // 2
    ctx.close()


def test_fmt_star_random_polygons_and_truncation(oracle):
    import torch
    import sea_current_amd as sc
    ctx = sc.Context(0)
    rng = np.random.default_rng(5)
    polys = []
    for _ in range(14):
        c = rng.uniform(-4, 4, 2); r = rng.uniform(0.3, 1.0); k = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        polys.append([(c[0] + r * np.cos(a), c[1] + r * np.sin(a)) for a in ang])
    lines, off = _world(polys)
    samples, _ = oracle.sample_free(600, (-5, 5, -5, 5), lines, off, hstate=(3, 8, 7, 27))
    starts = samples[rng.integers(1, 600, 40)]
    goals = samples[rng.integers(1, 600, 40)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    for Lmax in (64, 3):
        out = ctx.fmt_star(t(samples), t(starts), t(goals), 1.3, t(lines), Lmax=Lmax)
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in out.items()}
        for q in range(40):
            ref = oracle.fmt_star(samples, starts[q], goals[q], 1.3, lines, Lmax=Lmax)
            assert got["status"][q] == ref["status"] and got["len"][q] == ref["len"], (Lmax, q)
            if ref["status"] == 0:
                assert got["cost"][q] == np.float32(ref["cost"]) and np.array_equal(got["path"][q, :ref["len"]], ref["path"]), (Lmax, q)
    ctx.close()
