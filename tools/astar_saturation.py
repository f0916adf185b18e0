"""Scratch: tail-free A* throughput -- R copies of one long query in a single launch (every wave does the same work),
to find where the chip saturates (expansions/s vs resident waves) and to count PMC events per expansion."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
fam = sys.argv[1] if len(sys.argv) > 1 else "salt20"
occ = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()); torch.cuda.synchronize()
ex, pop, kc, stp = ctx.astar_debug_stats(1024)
j = int(np.argsort(ex)[len(ex) * 3 // 4])       # a fairly long query
for reps in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "64,512,1024,2048,4096".split(","))]:
    s1 = torch.from_numpy(np.repeat(s[j:j + 1], reps)).cuda(); g1 = torch.from_numpy(np.repeat(g[j:j + 1], reps)).cuda()
    for _ in range(2): ctx.astar_batch(d2, s1, g1)
    torch.cuda.synchronize(); t = time.perf_counter(); ctx.astar_batch(d2, s1, g1); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(fam, "x%d: %.2f ms, %d expansions / %d popped / %d steps each -> %.2f G expansions/s" % (reps, dt * 1e3, ex[j], pop[j], stp[j], reps * ex[j] / dt / 1e9), flush=True)
