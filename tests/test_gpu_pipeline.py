"""GPU parity of the batched post-planner sequence (sea_current_amd.pipeline.smooth_batch: from_path -> arclength ->
TOPP-RA along the arclength -> sampling -> resample(nudge) -> curvature; the reference's examples/zmq_test.cpp:66-93 for
a batch of paths) against the same sequence on the CPU restatement, path by path; and the recorded run of the reference
as a batch of one."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sea_current_amd as sc
    c = sc.Context(0)
    yield c
    c.close()


def test_recorded_run_as_a_batch_of_one(ctx, golden_dir):
    import torch
    from sea_current_amd import pipeline
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    wp = torch.from_numpy(fx["waypoints"][None].copy()).cuda()
    out = pipeline.smooth_batch(ctx, wp, vmax=float(fx["vel_lim"][1]), amax=float(fx["acc_lim"][1]), dt=0.02, N=100, max_len=4400)
    ctx.synchronize()
    L = int(out["length"][0])
    assert L == fx["pos"].shape[0] and int(out["resample_status"][0]) == 0
    pts = out["pts"].cpu().numpy()
    assert np.abs(pts[:, 0] - fx["pos_x"]).max() < 5e-5 and np.abs(pts[:, 1] - fx["pos_y"]).max() < 5e-5
    assert np.abs(out["ang_vel"].cpu().numpy() - fx["ang_vel"]).max() < 2e-6


def test_batch_of_astar_paths_matches_cpu_sequence(ctx, oracle):
    """256 A* paths of a 512^2 map -> 16 waypoints each -> the whole sequence in one batch; every path against the CPU
    sequence: arclength to 2e-6 relative, the same number of samples, positions / points to 2e-4 m, curvature where sane."""
    import torch
    from sea_current_amd import pipeline, synth
    occ = synth.block_grid(512, 512, 0.2, seed=5)
    d2 = ctx.edt(torch.from_numpy(occ).cuda())
    d2h = d2.cpu().numpy()
    s, g = synth.queries(d2h >= 4, 256)
    res = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), r2=4, Lmax=2048)
    ctx.synchronize()
    ln = res["len"].cpu().numpy()
    ok = (res["status"].cpu().numpy() == 0) & (ln >= 64)
    assert ok.sum() >= 150
    wp = pipeline.waypoints_from_cells(res["path"].cpu().numpy()[ok], ln[ok], 512, n_wp=16, cell_m=0.05)
    out = pipeline.smooth_batch(ctx, torch.from_numpy(wp).cuda(), vmax=1.0, amax=0.5, dt=0.02, N=100)
    ctx.synchronize()
    got = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in out.items()}
    off = got["offsets"]
    same_len = 0
    for b in range(wp.shape[0]):
        ref = oracle.smooth_one(wp[b], vmax=1.0, amax=0.5, dt=0.02, N=100)
        assert got["toppra_status"][b] == ref["toppra_status"] == 0, b
        assert abs(float(got["arclength"][b]) - float(ref["arclength"])) <= 2e-6 * float(ref["arclength"]), b
        assert np.abs(got["ctrl"][15 * b:15 * (b + 1)] - ref["ctrl"]).max() < 1e-5, b
        L = int(got["length"][b])
        assert abs(L - ref["length"]) <= 1, b           # an arclength one ulp apart can move ceil(T / dt) by one
        if L != ref["length"]:
            continue
        same_len += 1
        sl = slice(off[b], off[b + 1])
        assert got["resample_status"][b] == ref["status"] == 0, b
        assert np.abs(got["pos"][sl] - ref["pos"]).max() < 2e-4, b
        assert np.abs(got["vel"][sl] - ref["vel"]).max() < 2e-5, b
        assert np.abs(got["pts"][sl] - ref["pts"]).max() < 2e-4, b
        k = np.abs(ref["curvature"]) < 1e2
        assert np.allclose(got["curvature"][sl][k], ref["curvature"][k], rtol=2e-3, atol=2e-3), b
    assert same_len >= 0.95 * wp.shape[0]
