"""Scratch: one TOPP-RA batch (1024 plans, 6 joints, 200 stages) a few times, for PMC counting."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
pl = synth.toppra_plans(1024, dof=6)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
targs = (tt(pl["p0"]), tt(pl["p1"]), tt(pl["v0"]), tt(pl["v1"]), tt(-pl["vlim"]), tt(pl["vlim"]), tt(-pl["alim"]), tt(pl["alim"]))
for _ in range(4):
    tp = ctx.toppra(*targs, N=200)
    smp = ctx.toppra_sample(targs[0], targs[1], targs[2], targs[3], tp["x"], tp["t"], 0.02, 512)
torch.cuda.synchronize()
print("ok", int((tp["status"] == 0).sum()))
