"""GPU parity of the legal-move bytes (sc_moves_i32_u8) vs the oracle: the three forms of the kernel (four rows x four cells
per thread, four cells, one cell) on shapes that select each, clearance radii, stacked grids through the A* entry point."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(1024, 1024), (64, 32), (260, 30), (100, 37), (33, 5), (4, 4), (1, 1), (2052, 8)])
@pytest.mark.parametrize("r2", [0, 1, 4, 9])
def test_moves_match_oracle(oracle, W, H, r2):
    import torch
    import sea_current_amd as sc
    ctx = sc.Context(0)
    rng = np.random.default_rng(W * 31 + H + r2)
    occ = (rng.random((H, W)) < 0.15).astype(np.uint8)
    d2 = oracle.edt(occ)
    got = ctx.moves(torch.from_numpy(d2).cuda(), r2)
    ctx.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.moves(d2, r2))
    ctx.close()
