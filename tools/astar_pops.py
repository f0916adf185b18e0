"""Scratch: popped entries vs valid expansions (stale fraction) -- needs a library built with -DASTAR_COUNT_POPS."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
os.environ["SC_ASTAR_DEBUG"] = "1"
import numpy as np, torch
import sea_current_amd as sc
sc.LIB_PATH = os.path.join(sc.NATIVE_DIR, sys.argv[1])
from sea_current_amd import synth
ctx = sc.Context(0)
for fam in ("salt05", "salt20", "blocks"):
    occ = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
    d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
    ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()); torch.cuda.synchronize()
    ex, it = ctx.astar_debug_stats(1024)
    print(fam, "expansions %d, popped %d -> stale %.1f %%, steps %d" % (ex.sum(), it[:, 1].sum(), 100.0 * (it[:, 1].sum() - ex.sum()) / it[:, 1].sum(), it[:, 0].sum()))
