import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
os.environ["SC_ASTAR_DEBUG"] = "1"
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
sc.LIB_PATH = os.path.join(ROOT, "sea-current_amd", sys.argv[1] if len(sys.argv) > 1 else "libsc_stamps.so")
ctx = sc.Context(0)
occ = synth.salt_grid(1024, 1024, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()); torch.cuda.synchronize()
for j in (5, 17):
    s1 = torch.from_numpy(s[j:j + 1].copy()).cuda(); g1 = torch.from_numpy(g[j:j + 1].copy()).cuda()
    out = ctx.astar_batch(d2, s1, g1)
    torch.cuda.synchronize()
    ex, it = ctx.astar_debug_stats(1)
    n = int(out["status"][0])
    print("query", j, "iters", n, "ticks/iter: pop %.0f  mem %.0f  rest %.0f  | whole search %.0f" % (it[0, 0] * 1024.0 / n, it[0, 1] * 1024.0 / n, ex[0] * 1024.0 / n, int(out["cost"][0]) * 1024.0 / n),
          "| wide steps %d (%.0f ticks each), steps fed from HBM %d" % (int(out["len"][0]), int(out["path"][0, 0]) * 1024.0 / max(int(out["len"][0]), 1), int(out["path"][0, 1])))
