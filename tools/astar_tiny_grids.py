"""One-off: grids of 1..11 cells a side, end points anywhere (also outside the grid): EDT and A* against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
rng = np.random.default_rng(3)
n = 0
for r in range(400):
    W, H = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    occ = (rng.random((H, W)) < rng.choice([0.0, 0.1, 0.3, 0.6])).astype(np.uint8)
    d2g = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    d2 = oracle.edt(occ)
    assert np.array_equal(d2g.cpu().numpy(), d2), (r, W, H)
    Q = 40
    s = rng.integers(-1, W * H + 1, Q).astype(np.int32); g = rng.integers(-1, W * H + 1, Q).astype(np.int32)
    ref = oracle.astar_batch(d2, s, g, Lmax=64)
    out = ctx.astar_batch(d2g, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=64); torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for k in ("status", "cost", "len"):
        assert np.array_equal(got[k], ref[k]), (r, W, H, k, got[k], ref[k])
    for q in range(Q):
        if ref["status"][q] == 0:
            assert np.array_equal(got["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]), (r, q)
    n += Q
print("tiny grids ok:", n, "queries")
