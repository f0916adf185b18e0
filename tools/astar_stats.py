"""Scratch: per-query A* statistics (expansions, sub-iterations, cycles) on the bench maps."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
os.environ["SC_ASTAR_DEBUG"] = "1"
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
for fam in sys.argv[1].split(","):
    occ = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
    d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
    sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
    for _ in range(2): out = ctx.astar_batch(d2, sd, gd)
    torch.cuda.synchronize(); t = time.perf_counter(); out = ctx.astar_batch(d2, sd, gd); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ex, it = ctx.astar_debug_stats(1024)
    kc = it[:, 1].astype(np.float64); ni = it[:, 0].astype(np.float64)
    print(fam, "batch ms %.2f" % (dt * 1e3), "| exp mean %d max %d | iters mean %d max %d | nodes/iter mean %.1f | kcycles mean %d max %d -> us/iter mean %.2f, slowest query %.2f (at 100 MHz memtime: x10ns)" % (
        ex.mean(), ex.max(), ni.mean(), ni.max(), ex.sum() / ni.sum(), kc.mean(), kc.max(), (kc * 1024 / 100.0 / np.maximum(ni, 1)).mean(), (kc.max() * 1024 / 100.0 / ni[kc.argmax()])))

# single-query latency: the slowest query of the last family, alone on the GPU
j = int(kc.argmax())
for reps in (1, 16, 128):
    s1 = torch.from_numpy(np.repeat(s[j:j + 1], reps)).cuda(); g1 = torch.from_numpy(np.repeat(g[j:j + 1], reps)).cuda()
    for _ in range(2): ctx.astar_batch(d2, s1, g1)
    torch.cuda.synchronize(); t = time.perf_counter(); ctx.astar_batch(d2, s1, g1); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ex1, it1 = ctx.astar_debug_stats(reps)
    print("slowest query x%d alone: %.2f ms, %d iters, %.0f cycles/iter, %d expansions" % (reps, dt * 1e3, it1[0, 0], it1[0, 1] * 1024.0 / it1[0, 0], ex1[0]))
