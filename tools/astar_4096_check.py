"""One-off: BASELINE configs[3]'s shape at size -- 8192 queries on a 4096^2 grid through the batch kernel, the first 768
of them (and the 64 most expensive) against the oracle: status, cost, length, path, expansion count."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
occ = synth.salt_grid(4096, 4096, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
d2h = d2.cpu().numpy()
s, g = synth.queries(d2h >= 1, 8192)
out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=16384); torch.cuda.synchronize()
ex = ctx.astar_debug_stats(8192)[0]
got = {k: v.cpu().numpy() for k, v in out.items()}
sel = np.unique(np.concatenate([np.arange(768), np.argsort(-ex)[:64]]))
t = time.time()
ref = oracle.astar_batch(d2h, s[sel], g[sel], Lmax=16384, nthreads=16)
print("oracle: %d queries in %.1f s" % (sel.shape[0], time.time() - t), flush=True)
for k in ("status", "cost", "len"):
    assert np.array_equal(got[k][sel], ref[k]), k
assert np.array_equal(ex[sel], ref["expanded"])
for i, q in enumerate(sel):
    if ref["status"][i] == 0:
        assert np.array_equal(got["path"][q, :ref["len"][i]], ref["path"][i, :ref["len"][i]]), q
print("4096^2: %d of 8192 queries equal to the oracle (paths up to %d cells, up to %d expansions)" % (sel.shape[0], ref["len"].max(), ref["expanded"].max()))
