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
