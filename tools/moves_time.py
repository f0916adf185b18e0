"""Scratch: time of the legal-move kernel alone on G stacked 1024^2 grids.  python tools/moves_time.py [G]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python"))
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = sc.Context(0)
occ = torch.from_numpy(np.stack([synth.salt_grid(1024, 1024, 0.2, seed=5 + i) for i in range(G)])).cuda()
d2 = ctx.edt(occ)
ctx.synchronize()
s = torch.zeros(1024, dtype=torch.int32, device="cuda"); g = torch.full((1024,), 5, dtype=torch.int32, device="cuda")
qg = torch.zeros(1024, dtype=torch.int32, device="cuda")
ctx.set_timing(True); ctx.reset_timing()
for _ in range(10):
    ctx.astar_batch_multi(d2, qg, s, g, Lmax=64)      # trivial queries: the launch is the moves kernel + an empty search
ctx.synchronize()
ms, n = ctx.get_timing(sc.K_MOVES)
print("moves kernel, %d x 1024^2: %.1f us per launch (%d launches) = %.0f GB/s of 5 B/cell" % (G, ms / n * 1e3, n, 5 * G * 1024 * 1024 / (ms / n * 1e-3) / 1e9))
