"""Scratch: TOPP-RA sweep + sampling time against the batch size (6 joints, 200 stages)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
for P in (1024, 8192, 65536):
    pl = synth.toppra_plans(P, dof=6)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    targs = (tt(pl["p0"]), tt(pl["p1"]), tt(pl["v0"]), tt(pl["v1"]), tt(-pl["vlim"]), tt(pl["vlim"]), tt(-pl["alim"]), tt(pl["alim"]))
    for _ in range(2):
        tp = ctx.toppra(*targs, N=200); smp = ctx.toppra_sample(targs[0], targs[1], targs[2], targs[3], tp["x"], tp["t"], 0.02, 512)
    torch.cuda.synchronize(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(5):
        tp = ctx.toppra(*targs, N=200); smp = ctx.toppra_sample(targs[0], targs[1], targs[2], targs[3], tp["x"], tp["t"], 0.02, 512)
    torch.cuda.synchronize()
    a, _ = ctx.get_timing(sc.K_TOPPRA); b, _ = ctx.get_timing(sc.K_TOPPRA_SAMPLE); ctx.set_timing(False)
    print("P", P, "sweep ms %.3f sample ms %.3f -> %.2f M plans/s; ok %d" % (a / 5, b / 5, P / ((a + b) / 5 * 1e-3) / 1e6, int((tp["status"] == 0).sum())), flush=True)
