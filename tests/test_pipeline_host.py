"""Host side of the batched smoothing sequence (no GPU): waypoints from grid paths, and that the CPU sequence it is checked
against accepts them (no NaN from the reference's tangent construction on collinear grid cells)."""
import numpy as np


def test_waypoints_from_cells_and_cpu_sequence(oracle):
    from sea_current_amd import pipeline
    W = 300
    # a straight column, a straight row, a diagonal and a staircase: the collinear cases of grid paths
    col = np.arange(200) * W + 7
    row = 40 * W + np.arange(200)
    dia = np.arange(200) * W + np.arange(200)
    stair = np.cumsum(np.where(np.arange(200) % 2 == 0, 1, W)) + 5
    L = 256
    cells = np.zeros((4, L), np.int32)
    for k, p in enumerate((col, row, dia, stair)):
        cells[k, :200] = p
    wp = pipeline.waypoints_from_cells(cells, np.full(4, 200, np.int32), W, n_wp=16, cell_m=0.05, jitter=0.2)
    assert wp.shape == (4, 16, 2) and wp.dtype == np.float32
    for k, p in enumerate((col, row, dia, stair)):
        assert np.allclose(wp[k, 0], [p[0] % W * 0.05, p[0] // W * 0.05]) and np.allclose(wp[k, -1], [p[199] % W * 0.05, p[199] // W * 0.05])
        idx = np.rint(np.linspace(0, 1, 16) * 199).astype(int)
        exact = np.stack([p[idx] % W, p[idx] // W], axis=1) * 0.05
        assert np.abs(wp[k] - exact).max() <= 0.2 * 0.05 + 1e-6                  # moved by at most `jitter` cells
        ctrl = oracle.bezier_from_path(wp[k])
        assert np.isfinite(ctrl).all()
        r = oracle.smooth_one(wp[k])
        assert r["status"] == 0 and r["toppra_status"] == 0 and np.isfinite(r["pts"]).all() and r["length"] > 100
    # without the jitter the straight paths are what the reference's acos cannot take
    flat = pipeline.waypoints_from_cells(cells[:1], np.full(1, 200, np.int32), W, n_wp=16, cell_m=0.05, jitter=0.0)
    assert np.array_equal(flat[0, :, 0], np.full(16, np.float32(7 * 0.05)))
