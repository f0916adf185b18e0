"""Randomised parity rounds (HIP path through the C ABI against the CPU oracle) shared by the `-m gpu` suite
(tests/test_gpu_stress.py: bounded to seconds) and the long runs of tools/*_stress.py.  Every function draws one case from
`rng`, runs it on `ctx` and asserts bit-exact equality (EDT, A*: statuses, costs, lengths, paths AND the number of nodes every
search expanded) or the stated tolerance (TOPP-RA); it returns what it covered."""
import numpy as np


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def astar_round(ctx, oracle, rng, Q=96, max_side=400, nthreads=8):
    """One random map (salt / blocks / walls with gaps), random clearance, Q random queries between traversable cells."""
    import torch
    from sea_current_amd import synth
    W, H = int(rng.integers(9, max_side)), int(rng.integers(9, max_side))
    fam = int(rng.integers(0, 3))
    if fam == 0:
        occ = synth.salt_grid(W, H, float(rng.uniform(0.02, 0.45)), seed=int(rng.integers(1 << 30)))
    elif fam == 1:
        occ = synth.block_grid(W, H, float(rng.uniform(0.05, 0.4)), seed=int(rng.integers(1 << 30)), smin=2, smax=max(3, min(W, H) // 4))
    else:  # maze-like: salt + walls with gaps
        occ = synth.salt_grid(W, H, 0.05, seed=int(rng.integers(1 << 30)))
        for x in range(4, W - 4, int(rng.integers(5, 17))):
            occ[1:H - 1, x] = 1
            for _ in range(2):
                y = int(rng.integers(1, H - 1)); occ[max(1, y - 1):y + 2, x] = 0
    r2 = int(rng.choice([0, 0, 1, 2, 4, 9]))
    d2 = oracle.edt(occ)
    trav = d2 >= max(r2, 1)
    if trav.sum() < 4:
        return 0
    free = np.flatnonzero(trav.ravel()).astype(np.int32)
    s = rng.choice(free, Q).astype(np.int32); g = rng.choice(free, Q).astype(np.int32)
    Lmax = 4 * (W + H)
    ref = oracle.astar_batch(d2, s, g, r2=r2, Lmax=Lmax, nthreads=nthreads)
    out = ctx.astar_batch(_t(d2), _t(s), _t(g), r2=r2, Lmax=Lmax)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    tag = (W, H, fam, r2)
    for k in ("status", "cost", "len"):
        assert np.array_equal(got[k], ref[k]), (tag, k)
    for q in range(Q):
        if ref["status"][q] == 0:
            assert np.array_equal(got["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]), (tag, q)
    ex = ctx.astar_debug_stats(Q)[0]
    bad = np.flatnonzero(ex != ref["expanded"])
    assert bad.size == 0, ("expansion counts differ", tag, bad[:8].tolist(), ex[bad[:8]].tolist(), ref["expanded"][bad[:8]].tolist())
    return Q


def tiny_grid_round(ctx, oracle, rng, Q=40):
    """Grids of 1 .. 11 cells a side, end points anywhere (also outside the grid): EDT and A*."""
    import torch
    W, H = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    occ = (rng.random((H, W)) < rng.choice([0.0, 0.1, 0.3, 0.6])).astype(np.uint8)
    d2g = ctx.edt(_t(occ)); torch.cuda.synchronize()
    d2 = oracle.edt(occ)
    assert np.array_equal(d2g.cpu().numpy(), d2), (W, H)
    s = rng.integers(-1, W * H + 1, Q).astype(np.int32); g = rng.integers(-1, W * H + 1, Q).astype(np.int32)
    ref = oracle.astar_batch(d2, s, g, Lmax=64)
    out = ctx.astar_batch(d2g, _t(s), _t(g), Lmax=64); torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for k in ("status", "cost", "len"):
        assert np.array_equal(got[k], ref[k]), (W, H, k, got[k], ref[k])
    for q in range(Q):
        if ref["status"][q] == 0:
            assert np.array_equal(got["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]), (W, H, q)
    assert np.array_equal(ctx.astar_debug_stats(Q)[0], ref["expanded"]), (W, H)
    return Q


def edt_round(ctx, oracle, rng, max_cells=6_000_000):
    """Random shape (odd widths and heights, wide rows, very sparse to very dense grids, batches)."""
    import torch
    while True:
        W = int(rng.choice([rng.integers(1, 70), rng.integers(70, 1025), rng.integers(1025, 2700), rng.integers(2700, 6000), 1024, 512, 1023, 1025, 2048, 4096]))
        H = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), 32, 33, 31, 64]))
        B = int(rng.choice([1, 1, 2, 5]))
        if B * W * H <= max_cells:
            break
    p = float(rng.choice([0.0, 1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.2, 0.5, 0.95]))
    occ = (rng.random((B, H, W)) < p).astype(np.uint8)
    if rng.random() < 0.2 and H > 2 and W > 2:
        occ[:] = 0; occ[0, H // 2, W // 2] = 1          # a single obstacle: distances up to the grid diagonal
    d2 = ctx.edt(_t(occ)); torch.cuda.synchronize()
    ctx.synchronize()          # where a context adapts to maps of open space (and back): every state gets exercised
    got = d2.cpu().numpy()
    for b in range(B):
        ref = oracle.edt(occ[b])
        assert np.array_equal(got[b], ref), (W, H, B, p, b, np.argwhere(got[b] != ref)[:3])
    return B * W * H


def toppra_round(ctx, oracle, rng):
    """Random plans: 1 .. 16 joints, 1 .. 300 stages, position-dependent or constant velocity limits, non-zero boundary
    velocities, some infeasible: statuses equal; K / x / u / t within 1e-9 relative, sampled profiles within 1e-5."""
    import torch
    dof = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16]))
    N = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 31, 50, 100, 200, int(rng.integers(1, 300))]))
    P = int(rng.choice([1, 3, 17, 40]))
    p0 = rng.uniform(-2, 2, (P, dof)); p1 = p0 + rng.uniform(-3, 3, (P, dof))
    v0 = rng.uniform(-1.5, 1.5, (P, dof)); v1 = rng.uniform(-1.5, 1.5, (P, dof))
    if rng.random() < 0.3:
        v0[:] = 0; v1[:] = 0
    al = rng.uniform(0.3, 5.0, (P, dof))
    alo, ahi = -al, al * rng.uniform(0.5, 1.5, (P, dof))
    if rng.random() < 0.15:
        k = int(rng.integers(P)); alo[k, 0], ahi[k, 0] = 1.0, 0.5          # an empty acceleration interval
    per_stage = rng.random() < 0.4
    if per_stage:
        sg = np.arange(N + 1) / N
        vhi = rng.uniform(0.3, 2.0, (P, 1, dof)) * (0.6 + 0.4 * np.abs(np.sin(rng.uniform(1, 6) * sg))[None, :, None])
    else:
        vhi = rng.uniform(0.3, 3.0, (P, dof))
    sd0, sd1 = (float(rng.uniform(0, 0.8)), float(rng.uniform(0, 0.8))) if rng.random() < 0.5 else (0.0, 0.0)
    out = ctx.toppra(_t(p0), _t(p1), _t(v0), _t(v1), _t(-vhi), _t(vhi), _t(alo), _t(ahi), N=N, sd_start=sd0, sd_end=sd1)
    smp = ctx.toppra_sample(_t(p0), _t(p1), _t(v0), _t(v1), out["x"], out["t"], 0.05, 2048)
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    sm = {k: v.cpu().numpy() for k, v in smp.items()}
    nok, worst = 0, 0.0
    for p in range(P):
        ref = oracle.toppra(p0[p], p1[p], v0[p], v1[p], -vhi[p], vhi[p], alo[p], ahi[p], N=N, sd_start=sd0, sd_end=sd1)
        assert o["status"][p] == ref["status"], (p, dof, N, o["status"][p], ref["status"])
        if ref["status"] != 0:
            continue
        nok += 1
        for k in ("K", "x", "u", "t"):
            err = np.max(np.abs(o[k][p] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
            worst = max(worst, err)
            assert err < 1e-9, (p, dof, N, k, err)
        rs = oracle.toppra_sample(p0[p], p1[p], v0[p], v1[p], ref["x"], ref["t"], 0.05, 2048)
        n = min(rs["length"], 2048)
        if sm["length"][p] != rs["length"]:
            # ceil(T / dt) can differ when T / dt is within rounding of an integer
            assert abs(ref["t"][-1] / 0.05 - round(ref["t"][-1] / 0.05)) < 1e-6, (p, sm["length"][p], rs["length"])
            continue
        for k in ("pos", "vel", "acc"):
            scale = np.max(np.abs(rs[k])) + 1e-30
            assert np.max(np.abs(sm[k][p, :, :n] - rs[k][:, :n])) / scale < 1e-5, (p, dof, N, k)
    return P, nok, worst
