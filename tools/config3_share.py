"""Scratch: BASELINE configs[3], one GPU's block (8192 queries on a 4096^2 grid) against the scratch budget of the A* slots."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
fam = sys.argv[1] if len(sys.argv) > 1 else "salt20"
occ = synth.salt_grid(4096, 4096, 0.2) if fam == "salt20" else synth.block_grid(4096, 4096, 0.2)
occd = torch.from_numpy(occ).cuda()
for gb in sys.argv[2:]:
    os.environ["SC_ASTAR_SLOT_GB"] = gb
    ctx = sc.Context(0)
    d2 = ctx.edt(occd); torch.cuda.synchronize()
    s, g = synth.queries(d2.cpu().numpy() >= 1, 8192)
    sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
    t = time.perf_counter(); out = ctx.astar_batch(d2, sd, gd, Lmax=16384); torch.cuda.synchronize(); t_first = time.perf_counter() - t
    t = time.perf_counter(); out = ctx.astar_batch(ctx.edt(occd), sd, gd, Lmax=16384); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ex = ctx.astar_last_expansions()
    print("budget %s GiB: first call %.2f s, then %.1f ms = %.0f plans/s, %.2f G exp/s, scratch %.1f GiB, found %d" % (
        gb, t_first, dt * 1e3, 8192 / dt, ex / dt / 1e9, ctx.scratch_bytes() / 2**30, int((out["status"] == 0).sum())), flush=True)
    del out; ctx.close(); torch.cuda.empty_cache()
