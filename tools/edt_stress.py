"""One-off stress: HIP EDT vs the exact CPU oracle on random shapes (odd widths and heights, wide rows, very sparse and
very dense grids, batches); any differing cell is fatal.  python tools/edt_stress.py [rounds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(7)
cells = 0
for r in range(rounds):
    W = int(rng.choice([rng.integers(1, 70), rng.integers(70, 1025), rng.integers(1025, 2700), rng.integers(2700, 6000), 1024, 512, 1023, 1025, 2048, 4096]))
    H = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), 32, 33, 31, 64]))
    B = int(rng.choice([1, 1, 2, 5]))
    p = float(rng.choice([0.0, 1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.2, 0.5, 0.95]))
    occ = (rng.random((B, H, W)) < p).astype(np.uint8)
    if rng.random() < 0.2 and H > 2 and W > 2:
        occ[:] = 0; occ[0, H // 2, W // 2] = 1          # a single obstacle: distances up to the grid diagonal
    d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    got = d2.cpu().numpy()
    for b in range(B):
        ref = oracle.edt(occ[b])
        assert np.array_equal(got[b], ref), (r, W, H, B, p, b, np.argwhere(got[b] != ref)[:3])
    cells += B * W * H
    if r % 20 == 0:
        print("round", r, "ok", (W, H, B, p), flush=True)
print("stress ok:", cells, "cells")
