"""The bench's `smoothing` workload alone (for rocprofv3): the A* paths of one 1024^2 salt20 step -> 16 waypoints each ->
smooth_batch, 20 times.  python tools/smoothing_one.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python"))
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import pipeline, synth
ctx = sc.Context(0)
occ = synth.salt_grid(1024, 1024, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda())
s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
res = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=4096)
ctx.synchronize()
ln = res["len"].cpu().numpy()
ok = (res["status"].cpu().numpy() == 0) & (ln >= 64)
wp = torch.from_numpy(pipeline.waypoints_from_cells(res["path"].cpu().numpy()[ok], ln[ok], 1024)).cuda()
sm = pipeline.smooth_batch(ctx, wp)
ctx.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    sm = pipeline.smooth_batch(ctx, wp, max_len=sm["max_len"], nudge="nonudge" not in sys.argv)
ctx.synchronize()
print("paths %d samples %d: %.3f ms per batch" % (wp.shape[0], sm["pos"].shape[0], (time.perf_counter() - t0) / 20 * 1e3))
