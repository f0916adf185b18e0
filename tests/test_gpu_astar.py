"""GPU parity: batched HIP A* (through the C ABI) vs the CPU oracle -- bit-exact costs, g fields,
parent chains (paths), statuses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    import sea_current_amd as sc
    assert torch.cuda.is_available()
    c = sc.Context(0)
    yield c
    c.close()


def _run(ctx, d2, s, g, r2=0, Lmax=4096):
    import torch
    out = ctx.astar_batch(torch.from_numpy(d2).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), r2=r2, Lmax=Lmax)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    got["expanded"] = ctx.astar_debug_stats(s.shape[0])[0]      # nodes every search expanded
    return got


def _compare(gpu, ref, Q):
    assert np.array_equal(gpu["status"], ref["status"])
    assert np.array_equal(gpu["cost"], ref["cost"])
    assert np.array_equal(gpu["len"], ref["len"])
    if "expanded" in gpu:
        # the search expands exactly E = {n : g*(n) + h(n) <= C*}: a node expanded twice (a stale closed bit) or a pruned
        # one that should not have been shows here even when the path comes out right
        assert np.array_equal(gpu["expanded"], ref["expanded"]), np.flatnonzero(gpu["expanded"] != ref["expanded"])[:8]
    for q in range(Q):
        if ref["status"][q] == 0:
            L = ref["len"][q]
            assert np.array_equal(gpu["path"][q, :L], ref["path"][q, :L]), q


@pytest.mark.parametrize("W,H,p,r2,seed", [(40, 28, 0.1, 0, 1), (64, 64, 0.25, 0, 2), (97, 61, 0.33, 0, 3),
                                           (128, 128, 0.05, 4, 4), (256, 256, 0.2, 0, 5)])
def test_astar_matches_oracle(ctx, oracle, W, H, p, r2, seed):
    from sea_current_amd import synth
    occ = synth.salt_grid(W, H, p, seed=seed)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= max(r2, 1), 64, seed=seed)
    # add cross-component / invalid / trivial queries
    rng = np.random.default_rng(seed)
    free = np.flatnonzero((d2 >= max(r2, 1)).ravel()).astype(np.int32)
    s = np.concatenate([s, rng.choice(free, 16), [free[0], -1, W * H, np.flatnonzero(occ.ravel())[0]]]).astype(np.int32)
    g = np.concatenate([g, rng.choice(free, 16), [free[0], free[0], free[0], free[0]]]).astype(np.int32)
    Q = s.shape[0]
    ref = oracle.astar_batch(d2, s, g, r2=r2, Lmax=2048, nthreads=4)
    _compare(_run(ctx, d2, s, g, r2=r2, Lmax=2048), ref, Q)


def test_astar_gfield_bit_exact(ctx, oracle):
    import torch
    from sea_current_amd import synth
    occ = synth.block_grid(160, 120, 0.2, seed=3, smin=3, smax=24)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, 6, seed=11)
    d2g = torch.from_numpy(d2).cuda()
    for q in range(6):
        ref = oracle.astar(d2, s[q], g[q], want_g=True)
        gf, cost, status = ctx.astar_gfield(d2g, s[q], g[q])
        assert status == ref["status"] and cost == ref["cost"]
        # bit-exact on E (every node with g* + h <= C*: what paths and parents are read from); outside E the kernel's
        # successor pruning leaves upper bounds of the oracle's value
        H_, W_ = d2.shape
        yy, xx = np.divmod(np.arange(W_ * H_), W_)
        dx, dy = np.abs(xx - g[q] % W_), np.abs(yy - g[q] // W_)
        h = (10 * np.maximum(dx, dy) + 4 * np.minimum(dx, dy)).reshape(H_, W_)
        rg = ref["g"].astype(np.int64)
        inE = (ref["g"] != 0xFFFFFFFF) & (rg + h <= ref["cost"])
        assert inE.sum() == ref["expanded"]
        assert np.array_equal(gf[inE], ref["g"][inE])
        assert np.all(gf[~inE] >= ref["g"][~inE])
    # no path: g* over the whole component
    occ = np.zeros((32, 48), np.uint8); occ[:, 20] = 1
    d2 = oracle.edt(occ)
    ref = oracle.astar(d2, 0, 47, want_g=True)
    gf, cost, status = ctx.astar_gfield(torch.from_numpy(d2).cuda(), 0, 47)
    assert status == 1 and cost == -1 and np.array_equal(gf, ref["g"])


def test_astar_truncated_and_lmax(ctx, oracle):
    occ = np.zeros((3, 200), np.uint8)
    d2 = np.full((3, 200), 9, np.int32); d2[0] = 0; d2[2] = 0   # corridor
    s = np.array([200], np.int32); g = np.array([399], np.int32)
    ref = oracle.astar_batch(d2, s, g, Lmax=50)
    out = _run(ctx, d2, s, g, Lmax=50)
    assert out["status"][0] == 3 == ref["status"][0] and out["len"][0] == 200 == ref["len"][0]
    out = _run(ctx, d2, s, g, Lmax=200)
    assert out["status"][0] == 0 and np.array_equal(out["path"][0], np.arange(200, 400))


def test_astar_1024_bench_config(ctx, oracle):
    """BASELINE configs[1] at full size on all three synthetic maps (first 96 queries vs the oracle)."""
    from sea_current_amd import synth
    for occ in (synth.salt_grid(1024, 1024, 0.05), synth.salt_grid(1024, 1024, 0.20), synth.block_grid(1024, 1024, 0.2)):
        d2 = oracle.edt(occ)
        s, g = synth.queries(d2 >= 1, 96)
        ref = oracle.astar_batch(d2, s, g, Lmax=4096, nthreads=8)
        _compare(_run(ctx, d2, s, g, Lmax=4096), ref, 96)


def test_astar_host_entry_point(ctx, oracle):
    from sea_current_amd import synth
    occ = synth.salt_grid(96, 96, 0.2, seed=8)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, 20, seed=8)
    ref = oracle.astar_batch(d2, s, g, Lmax=512)
    _compare(ctx.astar_batch_host(d2, s, g, Lmax=512), ref, 20)


def test_astar_4096_config3(ctx, oracle):
    """BASELINE configs[3] shape (4096^2 grid; one rank's slice of the 64k queries, cut to what the oracle checks in
    seconds): GPU EDT + A* vs the oracle, bit-exact, and the query generator's rank-independent indexing."""
    import torch
    from sea_current_amd import synth
    occ = synth.salt_grid(4096, 4096, 0.20)
    d2 = ctx.edt(torch.from_numpy(occ).cuda())
    torch.cuda.synchronize()
    d2h = d2.cpu().numpy()
    assert np.array_equal(d2h, oracle.edt(occ))
    trav = d2h >= 1
    s, g = synth.queries(trav, 64, first=8192 * 3)              # the first queries of rank 3 of 8
    out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=16384)
    torch.cuda.synchronize()
    ref = oracle.astar_batch(d2h, s, g, Lmax=16384, nthreads=16)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    got["expanded"] = ctx.astar_debug_stats(64)[0]
    _compare(got, ref, 64)
    assert (ref["status"] == 0).all() and ref["len"].max() > 2048


def test_astar_multi_grid_launch(ctx, oracle):
    """sc_astar_batch_multi: three different grids (and clearances are per call) in one launch, every query against the
    oracle on its own grid."""
    import torch
    from sea_current_amd import synth
    W = H = 192
    occs = [synth.salt_grid(W, H, 0.2, seed=21), synth.block_grid(W, H, 0.2, seed=22, smin=3, smax=20), synth.salt_grid(W, H, 0.05, seed=23)]
    d2s = [oracle.edt(o) for o in occs]
    ss, gs, qg = [], [], []
    for k, d2 in enumerate(d2s):
        s, g = synth.queries(d2 >= 1, 40 + 8 * k, seed=30 + k)
        ss.append(s); gs.append(g); qg.append(np.full(s.shape[0], k, np.int32))
    s, g, qgrid = np.concatenate(ss), np.concatenate(gs), np.concatenate(qg)
    perm = np.random.default_rng(1).permutation(s.shape[0])          # queries of the grids interleaved
    s, g, qgrid = s[perm], g[perm], qgrid[perm]
    out = ctx.astar_batch_multi(torch.from_numpy(np.stack(d2s)).cuda(), torch.from_numpy(qgrid).cuda(), torch.from_numpy(s).cuda(),
                                torch.from_numpy(g).cuda(), Lmax=2048)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    got["expanded"] = ctx.astar_debug_stats(s.shape[0])[0]
    for k, d2 in enumerate(d2s):
        sel = np.flatnonzero(qgrid == k)
        ref = oracle.astar_batch(d2, s[sel], g[sel], Lmax=2048, nthreads=4)
        _compare({kk: vv[sel] for kk, vv in got.items()}, ref, sel.shape[0])


def test_astar_ring_overflow_retry(ctx, oracle, monkeypatch):
    """Bucket rings far too small for the map (SC_ASTAR_CAP = 1024 entries per level): the first launch gives up on the
    wide searches, the device-side retry launch (16x the ring space, enqueued behind it without a host round trip)
    finishes them.  Every query that reports SC_Q_OK must equal the oracle; what even the retry cannot hold is reported
    as SC_Q_RING_OVERFLOW, never as a wrong path."""
    import ctypes as C
    import torch
    import sea_current_amd as sc
    from sea_current_amd import synth
    occ = synth.salt_grid(768, 768, 0.05, seed=5)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, 96, seed=6)
    ref = oracle.astar_batch(d2, s, g, Lmax=4096, nthreads=8)
    monkeypatch.setenv("SC_ASTAR_CAP", "1024")
    got = _run(ctx, d2, s, g, Lmax=4096)
    buf = (C.c_int32 * 16)()
    ctx._l.sc_astar_debug_peek(ctx._h, buf)
    assert buf[1] > 0, "the small rings were expected to overflow in the first launch"
    ok = got["status"] == 0
    assert ok.sum() > 0 and set(np.unique(got["status"])) <= {0, sc.Q_RING_OVERFLOW}
    assert (ok.sum() > (96 - buf[1])), "the retry launch finished none of the overflowed queries"
    assert np.array_equal(got["cost"][ok], ref["cost"][ok]) and np.array_equal(got["len"][ok], ref["len"][ok])
    for q in np.flatnonzero(ok):
        assert np.array_equal(got["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]), q
    # with the default rings everything is found, first launch
    monkeypatch.delenv("SC_ASTAR_CAP")
    _compare(_run(ctx, d2, s, g, Lmax=4096), ref, 96)


def test_astar_full_headline_batch(ctx, oracle):
    """The whole headline batch (BASELINE configs[1]: 1024 queries on the 1024^2 salt20 grid, the bench's own inputs), every
    query against the oracle: status, cost, length, path, and the number of expanded nodes."""
    import torch
    from sea_current_amd import synth
    occ = synth.salt_grid(1024, 1024, 0.20)
    d2 = ctx.edt(torch.from_numpy(occ).cuda())
    torch.cuda.synchronize()
    d2h = d2.cpu().numpy()
    assert np.array_equal(d2h, oracle.edt(occ))
    s, g = synth.queries(d2h >= 1, 1024)
    out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=4096)
    torch.cuda.synchronize()
    ex = ctx.astar_debug_stats(1024)[0]
    ref = oracle.astar_batch(d2h, s, g, Lmax=4096, nthreads=16)
    _compare({k: v.cpu().numpy() for k, v in out.items()}, ref, 1024)
    assert np.array_equal(ex, ref["expanded"])
    # the same batch through the throughput build of the two-wavefront kernel (what batches larger than the chip run)
    import os
    import sea_current_amd as sc
    os.environ["SC_ASTAR_LATENCY"] = "0"
    try:
        c2 = sc.Context(0)
        out2 = c2.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=4096)
        c2.synchronize()
        ex2 = c2.astar_debug_stats(1024)[0]
        _compare({k: v.cpu().numpy() for k, v in out2.items()}, ref, 1024)
        assert np.array_equal(ex2, ref["expanded"])
        c2.close()
    finally:
        del os.environ["SC_ASTAR_LATENCY"]


def test_astar_scratch_fits_what_the_device_has_free(oracle):
    """The per-search scratch is budgeted per context but allocated on a shared device: with most of the HBM taken by
    somebody else a large batch must still run -- on fewer resident searches -- and give the same results."""
    import torch
    import sea_current_amd as sc
    from sea_current_amd import synth
    occ = synth.salt_grid(2048, 2048, 0.2, seed=12)
    Q = 2048
    c = sc.Context(0)
    try:
        d2 = c.edt(torch.from_numpy(occ).cuda())
        torch.cuda.synchronize()
        d2h = d2.cpu().numpy()
        s, g = synth.queries(d2h >= 1, Q, seed=12)
        sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
        free, _ = torch.cuda.mem_get_info()
        hog = torch.empty(max(0, free - (6 << 30)), dtype=torch.uint8, device="cuda")      # leave about 6 GiB: a fraction of the 17 GiB 2048 slots want
        out = c.astar_batch(d2, sd, gd, Lmax=8192)
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in out.items()}
        got_ex = c.astar_debug_stats(Q)[0]
        assert c.scratch_bytes() < (6 << 30)
        del hog
        torch.cuda.empty_cache()
    finally:
        c.close()
    assert (got["status"] == 0).all()
    sel = np.arange(0, Q, 32)
    ref = oracle.astar_batch(d2h, s[sel], g[sel], Lmax=8192, nthreads=16)
    sub = {k: v[sel] for k, v in got.items()}
    sub["expanded"] = got_ex[sel]
    _compare(sub, ref, sel.shape[0])
