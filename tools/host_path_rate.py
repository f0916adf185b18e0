"""Scratch: plans/s through the _host entry points (grid and queries in host memory, results copied back): the
PCIe-inclusive rate DESIGN.md quotes next to `value`."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
occ = synth.salt_grid(1024, 1024, 0.2)
d2 = ctx.edt_host(occ)
s, g = synth.queries(d2 >= 1, 1024)
for _ in range(2):
    out = ctx.astar_batch_host(ctx.edt_host(occ), s, g, Lmax=4096)
t = time.perf_counter()
n = 5
for _ in range(n):
    out = ctx.astar_batch_host(ctx.edt_host(occ), s, g, Lmax=4096)
dt = (time.perf_counter() - t) / n
print("host path: %.2f ms per step (EDT + 1024 queries, H2D of grid/queries, D2H of d2 and paths) -> %.0f plans/s" % (dt * 1e3, 1024 / dt))
