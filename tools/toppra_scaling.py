"""Scratch: TOPP-RA sweep time vs batch size (is it per-plan latency or throughput?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
for P in (1, 64, 1024, 4096, 16384):
    pl = synth.toppra_plans(P, dof=6)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (tt(pl["p0"]), tt(pl["p1"]), tt(pl["v0"]), tt(pl["v1"]), tt(-pl["vlim"]), tt(pl["vlim"]), tt(-pl["alim"]), tt(pl["alim"]))
    for N in (100, 200):
        for _ in range(2): ctx.toppra(*args, N=N)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ctx.toppra(*args, N=N)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print("P %5d N %3d: %.3f ms -> %.2f us per stage-step, %.0f plans/s" % (P, N, dt * 1e3, dt * 1e6 / (2 * N), P / dt), flush=True)
