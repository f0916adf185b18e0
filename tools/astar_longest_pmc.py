"""Scratch: the longest search of the headline batch ALONE on the chip (what the end of a one-call batch looks like), repeated,
for rocprofv3 --pmc: instructions and cycles per frontier step of the two wavefronts of one query."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
occ = synth.salt_grid(1024, 1024, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()); torch.cuda.synchronize()
ex, pop, kc, stp = ctx.astar_debug_stats(1024)
j = int(np.argmax(kc))
s1 = torch.from_numpy(s[j:j + 1].copy()).cuda(); g1 = torch.from_numpy(g[j:j + 1].copy()).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(2): ctx.astar_batch(d2, s1, g1)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): ctx.astar_batch(d2, s1, g1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
e1, p1, k1, st1 = ctx.astar_debug_stats(1)
print(json.dumps({"query": j, "ms_alone": dt * 1e3, "expansions": int(e1[0]), "popped": int(p1[0]), "kilocycles": int(k1[0]), "steps": int(st1[0]),
                  "cycles_per_step": 1024.0 * k1[0] / st1[0], "launches_of_the_dual_kernel_with_this_query_alone": reps + 2}))
