"""GPU parity: batched Bezier smoothing / evaluation / arclength (through the C ABI) vs the CPU oracle and vs
the reference's recorded arclength tables.  Tolerances: float32 outputs, 1e-6 relative."""
import os

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


def test_fixture_arclength_tables(ctx, oracle, golden_dir):
    import torch
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    path = torch.from_numpy(fx["waypoints"][None].copy()).cuda()
    ctrl = ctx.bezier_from_path(path, torch.tensor([3], dtype=torch.int32, device="cuda"))
    cum, seg_len = ctx.bezier_arclength(ctrl, 100)
    torch.cuda.synchronize()
    assert np.allclose(ctrl.cpu().numpy()[0], oracle.bezier_from_path(fx["waypoints"]), atol=1e-6)
    total = float(seg_len.sum())
    assert abs(total - float(fx["arclength"])) / total < 1e-6
    assert np.abs(cum.cpu().numpy() - fx["arclength_segments"]).max() < 5e-6


def test_batch_matches_oracle(ctx, oracle):
    import torch
    rng = np.random.default_rng(8)
    P, n_max = 17, 9
    npts = rng.integers(2, n_max + 1, P).astype(np.int32)
    path = np.cumsum(rng.uniform(0.3, 2.0, (P, n_max, 2)) * rng.choice([-1, 1], (P, n_max, 2)), axis=1).astype(np.float32)
    lines = rng.uniform(-5, 5, (6, 4)).astype(np.float32)
    for ln in (None, lines):
        ctrl = ctx.bezier_from_path(torch.from_numpy(path).cuda(), torch.from_numpy(npts).cuda(),
                                    lines=None if ln is None else torch.from_numpy(ln).cuda())
        cum, seg_len = ctx.bezier_arclength(ctrl, 50)
        torch.cuda.synchronize()
        ch, cumh, slh = ctrl.cpu().numpy(), cum.cpu().numpy().reshape(P, n_max - 1, 51), seg_len.cpu().numpy().reshape(P, n_max - 1)
        for p in range(P):
            ref = oracle.bezier_from_path(path[p, :npts[p]], lines=ln)
            assert np.allclose(ch[p, :npts[p] - 1], ref, rtol=1e-6, atol=1e-6), p
            assert np.all(ch[p, npts[p] - 1:] == 0)
            tot, rc = oracle.bezier_arclength(ch[p, :npts[p] - 1], 50)   # same float32 control points
            assert np.allclose(cumh[p, :npts[p] - 1], rc, rtol=2e-6, atol=1e-6), p
            assert abs(slh[p, :npts[p] - 1].sum() - tot) / tot < 2e-6


def test_eval_orders(ctx, oracle):
    import torch
    rng = np.random.default_rng(4)
    ctrl = rng.uniform(-3, 3, (11, 4, 2)).astype(np.float32)
    seg = rng.integers(0, 11, 500).astype(np.int32)
    t = rng.uniform(0, 1, 500).astype(np.float32)
    for order in (0, 1, 2):
        out = ctx.bezier_eval(torch.from_numpy(ctrl).cuda(), torch.from_numpy(seg).cuda(), torch.from_numpy(t).cuda(), order)
        torch.cuda.synchronize()
        ref = oracle.bezier_eval(ctrl, seg, t.astype(np.float64), order)
        assert np.allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-6), order


def test_resample_fixture_pipeline(ctx, oracle, golden_dir):
    """The whole recorded run of the reference (examples/zmq_test.cpp:66-93) on the GPU: waypoints -> from_path ->
    arclength -> TOPP-RA (1 dof along the arclength) -> sampling -> resample(nudge) -> angular velocity, against
    examples/output.json (pos_x / pos_y / ang_vel)."""
    import torch
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    dev = "cuda"
    ctrl = ctx.bezier_from_path(torch.from_numpy(fx["waypoints"][None].copy()).to(dev), torch.tensor([3], dtype=torch.int32, device=dev))
    cum, seg_len = ctx.bezier_arclength(ctrl, 100)
    AL = seg_len.sum().reshape(1)
    N = 100
    f64 = lambda v: torch.tensor([[v]], dtype=torch.float64, device=dev)
    vlo = torch.full((1, N + 1, 1), float(fx["vel_lim"][0]), dtype=torch.float64, device=dev)
    vhi = torch.full((1, N + 1, 1), float(fx["vel_lim"][1]), dtype=torch.float64, device=dev)
    p0, p1, v0, v1 = f64(0.0), AL.double().reshape(1, 1), f64(0.0), f64(0.0)
    res = ctx.toppra(p0, p1, v0, v1, vlo, vhi, f64(float(fx["acc_lim"][0])), f64(float(fx["acc_lim"][1])), N=N)
    smp = ctx.toppra_sample(p0, p1, v0, v1, res["x"], res["t"], float(np.float32(0.02)), max_len=4400)
    L = int(smp["length"][0])
    assert L == fx["pos"].shape[0]
    pos = smp["pos"][0, 0, :L].contiguous()
    off = torch.tensor([0, L], dtype=torch.int32, device=dev)
    out = ctx.bezier_resample(ctrl.reshape(-1, 4, 2), cum, AL, torch.tensor([0, 2], dtype=torch.int32, device=dev), pos, off, nudge=True)
    torch.cuda.synchronize()
    assert int(out["status"][0]) == 0
    pts = out["pts"].cpu().numpy()
    assert np.abs(pts[:, 0] - fx["pos_x"]).max() < 5e-5 and np.abs(pts[:, 1] - fx["pos_y"]).max() < 5e-5
    ang = (smp["vel"][0, 0, :L] * out["curvature"]).cpu().numpy()
    assert np.abs(ang - fx["ang_vel"]).max() < 2e-6
    # and from the recorded profile itself: same bar as the oracle
    pos2 = torch.from_numpy(fx["pos"].copy()).to(dev)
    out2 = ctx.bezier_resample(ctrl.reshape(-1, 4, 2), torch.from_numpy(fx["arclength_segments"].copy()).to(dev),
                               torch.tensor([float(fx["arclength"])], dtype=torch.float32, device=dev),
                               torch.tensor([0, 2], dtype=torch.int32, device=dev), pos2, off, nudge=True)
    torch.cuda.synchronize()
    ref = oracle.bezier_resample(ctrl.cpu().numpy().reshape(-1, 4, 2), fx["arclength_segments"], fx["arclength"], fx["pos"], True)
    assert np.array_equal(pos2.cpu().numpy(), ref["pos"]) and np.array_equal(out2["seg"].cpu().numpy(), ref["seg"])
    assert np.abs(out2["t"].cpu().numpy() - ref["t"]).max() < 2e-7
    assert np.abs(out2["pts"].cpu().numpy() - ref["pts"]).max() < 2e-6
    assert np.abs(out2["pts"].cpu().numpy()[:, 0] - fx["pos_x"]).max() < 2e-5
    assert np.abs(fx["vel"] * out2["curvature"].cpu().numpy() - fx["ang_vel"]).max() < 3e-7


def test_resample_batch_matches_oracle(ctx, oracle):
    """Ragged batch: 1..7 segments, glitchy profiles (the nudge's sequential replay path), with and without nudge."""
    import torch
    rng = np.random.default_rng(21)
    B, nsub = 23, 64
    ctrls, cums, als, pps, seg_off, prof_off = [], [], [], [], [0], [0]
    for b in range(B):
        nwp = int(rng.integers(2, 9))
        path = np.cumsum(rng.uniform(0.5, 3.0, (nwp, 2)) * rng.choice([-1, 1], (nwp, 2)), axis=0).astype(np.float32)
        c = oracle.bezier_from_path(path)
        tot, cum = oracle.bezier_arclength(c, nsub)
        cum = cum.astype(np.float32)
        AL = np.float32(cum[:, -1].sum(dtype=np.float32))
        n = int(rng.integers(40 * (nwp - 1), 400))
        pp = (np.sort(rng.uniform(0, 1, n)) * AL).astype(np.float32)
        if b % 3 != 2:                                           # leave some profiles clean
            k = rng.choice(np.arange(1, n - 1), max(1, n // 30), replace=False)
            pp[k] += rng.normal(0, 0.05 * AL, k.size).astype(np.float32)
        ctrls.append(c); cums.append(cum); als.append(AL); pps.append(pp)
        seg_off.append(seg_off[-1] + nwp - 1); prof_off.append(prof_off[-1] + n)
    dev = "cuda"
    tc = torch.from_numpy(np.concatenate(ctrls)).to(dev)
    tcum = torch.from_numpy(np.concatenate(cums)).to(dev)
    tal = torch.tensor(np.array(als), dtype=torch.float32, device=dev)
    tso = torch.tensor(seg_off, dtype=torch.int32, device=dev)
    tpo = torch.tensor(prof_off, dtype=torch.int32, device=dev)
    for nudge in (True, False):
        tpp = torch.from_numpy(np.concatenate(pps)).to(dev)
        out = ctx.bezier_resample(tc, tcum, tal, tso, tpp, tpo, nudge=nudge)
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in out.items()}
        gpp = tpp.cpu().numpy()
        for b in range(B):
            if not nudge and b % 3 != 2:
                continue                                         # glitchy profile without the nudge: the reference asserts
            ref = oracle.bezier_resample(ctrls[b], cums[b], als[b], pps[b], nudge)
            sl = slice(prof_off[b], prof_off[b + 1])
            assert got["status"][b] == ref["status"], b
            assert np.array_equal(gpp[sl], ref["pos"]), b
            if ref["status"]:
                continue
            assert np.array_equal(got["seg"][sl], ref["seg"]), b
            assert np.abs(got["t"][sl] - ref["t"]).max() < 5e-7, b
            assert np.abs(got["pts"][sl] - ref["pts"]).max() < 1e-5, b
            k = np.abs(ref["curvature"]) < 1e3                   # cusps of random splines: compare where curvature is sane
            assert np.allclose(got["curvature"][sl][k], ref["curvature"][k], rtol=2e-4, atol=1e-5), b


def test_general_degree_curve_and_free_chebfit(ctx, oracle):
    """sc_bezier_curve_batch (any degree) and sc_chebfit_batch / sc_chebeval_batch (the free functions of the header,
    sea_current.hpp:700-763, 1109-1170) against the oracle; a fit with tens of thousands of rows as in examples/test.cpp:168."""
    import torch
    rng = np.random.default_rng(9)
    for deg in (1, 2, 3, 6, 15):
        ctrl = rng.uniform(-4, 4, (5, deg + 1, 2)).astype(np.float32)
        seg = rng.integers(0, 5, 300).astype(np.int32)
        t = rng.uniform(0, 1, 300).astype(np.float32)
        got = ctx.bezier_curve(torch.from_numpy(ctrl).cuda(), torch.from_numpy(seg).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        want = oracle.bezier_curve(ctrl, seg, t.astype(np.float64))
        assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    # three problems of very different sizes in one batch
    sizes = [7, 300, 20002]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = np.concatenate([np.sort(rng.uniform(-1, 5, n)) for n in sizes]).astype(np.float32)
    y = (np.cos(1.3 * x) + 0.05 * x ** 3).astype(np.float32)
    for degree in (3, 10):
        coef, xr = ctx.chebfit(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(off).cuda(), degree)
        yh = ctx.chebeval(torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda(), coef, xr).cpu().numpy()
        coef, xr = coef.cpu().numpy(), xr.cpu().numpy()
        for b, n in enumerate(sizes):
            sl = slice(off[b], off[b + 1])
            c_ref, xmin, xmax = oracle.chebfit(x[sl], y[sl], degree)
            assert xr[b, 0] == np.float32(xmin) and xr[b, 1] == np.float32(xmax)
            y_ref = oracle.chebeval(x[sl], c_ref, xmin, xmax)
            # coefficients of an ill-conditioned small problem (7 rows, 10 columns would be rank deficient: skipped) move; values do not
            if n > degree:
                assert np.abs(yh[sl] - y_ref).max() < 5e-5 * max(1.0, np.abs(y_ref).max()), (degree, n)
                if n >= 300:
                    assert np.abs(coef[b] - c_ref).max() < 1e-4 * max(1.0, np.abs(c_ref).max())


def test_shrink_tangent_batch_matches_oracle(ctx, oracle):
    """sc_bezier_shrink_tangent_batch(_host) (bezier_spline::shrink_tangent, sea_current.hpp:575-596) against the oracle on
    random tangents, waypoints and walls, and against the hand-made cases of tests/test_oracle_bezier.py."""
    import ctypes as C
    import sea_current_amd as sc
    rng = np.random.default_rng(5)
    M, E = 500, 23
    T = rng.uniform(-3, 3, (M, 2)).astype(np.float32)
    Wp = rng.uniform(-5, 5, (M, 2)).astype(np.float32)
    lines = rng.uniform(-6, 6, (E, 4)).astype(np.float32)
    out = np.zeros((M, 2), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for k in (1.0, 0.4):
        st = sc.lib().sc_bezier_shrink_tangent_batch_host(ctx._h, p(T), p(Wp), M, C.c_float(k), p(lines), E, p(out))
        assert st == 0
        ref = oracle.bezier_shrink_tangent(T, Wp, k, lines)
        assert np.allclose(out, ref, rtol=1e-6, atol=1e-6)
        assert (np.abs(ref - k * T).max(axis=1) > 1e-3).sum() > 20          # a good share of the tangents is actually cut
    st = sc.lib().sc_bezier_shrink_tangent_batch_host(ctx._h, p(T), p(Wp), M, C.c_float(0.5), None, 0, p(out))
    assert st == 0 and np.allclose(out, 0.5 * T)
    one = np.zeros((1, 2), np.float32)
    wall = np.array([[1.5, -1, 1.5, 1]], np.float32)
    sc.lib().sc_bezier_shrink_tangent_batch_host(ctx._h, p(np.array([[4, 0]], np.float32)), p(np.zeros((1, 2), np.float32)), 1, C.c_float(0.5), p(wall), 1, p(one))
    assert np.allclose(one, [[1.5, 0]])


@pytest.mark.parametrize("nsub", [5, 8, 9, 40, 63, 64, 127, 150, 255, 256, 300])
def test_resample_table_sizes(ctx, oracle, nsub):
    """The per-segment fit runs in registers for tables of 10 .. 256 rows (one wavefront per segment, 1 .. 4 rows per lane)
    and in LDS otherwise: every size class, and the sizes either side of each boundary, against the oracle."""
    import torch
    rng = np.random.default_rng(1000 + nsub)
    B = 5
    ctrls, cums, als, pps, seg_off, prof_off = [], [], [], [], [0], [0]
    for b in range(B):
        nwp = int(rng.integers(2, 7))
        path = np.cumsum(rng.uniform(0.5, 3.0, (nwp, 2)) * rng.choice([-1, 1], (nwp, 2)), axis=0).astype(np.float32)
        c = oracle.bezier_from_path(path)
        tot, cum = oracle.bezier_arclength(c, nsub)
        cum = cum.astype(np.float32)
        AL = np.float32(cum[:, -1].sum(dtype=np.float32))
        n = int(rng.integers(30 * (nwp - 1), 300))
        pp = (np.sort(rng.uniform(0, 1, n)) * AL).astype(np.float32)
        ctrls.append(c); cums.append(cum); als.append(AL); pps.append(pp)
        seg_off.append(seg_off[-1] + nwp - 1); prof_off.append(prof_off[-1] + n)
    dev = "cuda"
    tpp = torch.from_numpy(np.concatenate(pps)).to(dev)
    out = ctx.bezier_resample(torch.from_numpy(np.concatenate(ctrls)).to(dev), torch.from_numpy(np.concatenate(cums)).to(dev),
                              torch.tensor(np.array(als), dtype=torch.float32, device=dev), torch.tensor(seg_off, dtype=torch.int32, device=dev),
                              tpp, torch.tensor(prof_off, dtype=torch.int32, device=dev), nudge=True)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for b in range(B):
        ref = oracle.bezier_resample(ctrls[b], cums[b], als[b], pps[b], True)
        sl = slice(prof_off[b], prof_off[b + 1])
        assert got["status"][b] == ref["status"], b
        if ref["status"]:
            continue
        assert np.array_equal(got["seg"][sl], ref["seg"]), b
        assert np.abs(got["t"][sl] - ref["t"]).max() < 5e-7, b
        assert np.abs(got["pts"][sl] - ref["pts"]).max() < 1e-5, b
