"""Scratch: batch time and per-query statistics of the A* kernel on the bench maps; the slowest query alone."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import sea_current_amd as sc
from sea_current_amd import synth
ctx = sc.Context(0)
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for fam in sys.argv[1].split(","):
    occ = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
    d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    s, g = synth.queries(d2.cpu().numpy() >= 1, Q)
    sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
    for _ in range(2): out = ctx.astar_batch(d2, sd, gd)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out = ctx.astar_batch(d2, sd, gd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = min(ts)
    ex, pop, kc, stp = ctx.astar_debug_stats(Q)
    buf = (C.c_int32 * 16)(); ctx._l.sc_astar_debug_peek(ctx._h, buf)
    st = out["status"].cpu().numpy()
    print(fam, "Q %d batch ms %.2f (%.0f plans/s) | exp mean %d max %d total %.1f M -> %.2f G exp/s | popped/expanded %.2f | query ms mean %.2f max %.2f (2.4 GHz) | overflowed %d (retry %d) | status %s" % (
        Q, dt * 1e3, Q / dt, ex.mean(), ex.max(), ex.sum() / 1e6, ex.sum() / dt / 1e9, pop.sum() / max(ex.sum(), 1), kc.mean() * 1024 / 2.4e6, kc.max() * 1024 / 2.4e6,
        buf[1], buf[3], np.bincount(st, minlength=5).tolist()), flush=True)
    for k in np.argsort(-kc)[:4]:
        print("   slow query %d: %.2f ms, %d expansions, %d popped, %d steps -> %.0f cycles/step, %.1f nodes/step, h(start) %d cost %d" % (
            k, kc[k] * 1024 / 2.4e6, ex[k], pop[k], stp[k], kc[k] * 1024.0 / max(stp[k], 1), pop[k] / max(stp[k], 1),
            10 * max(abs(s[k] % 1024 - g[k] % 1024), abs(s[k] // 1024 - g[k] // 1024)) + 4 * min(abs(s[k] % 1024 - g[k] % 1024), abs(s[k] // 1024 - g[k] // 1024)),
            int(out["cost"][k])), flush=True)
    print("   all queries: %.0f cycles/step, %.1f nodes/step" % (kc.sum() * 1024.0 / stp.sum(), pop.sum() / stp.sum()), flush=True)
    j = int(ex.argmax())
    for reps in (1,):
        s1 = torch.from_numpy(np.repeat(s[j:j + 1], reps)).cuda(); g1 = torch.from_numpy(np.repeat(g[j:j + 1], reps)).cuda()
        for _ in range(2): ctx.astar_batch(d2, s1, g1)
        torch.cuda.synchronize(); t = time.perf_counter(); ctx.astar_batch(d2, s1, g1); torch.cuda.synchronize(); dt1 = time.perf_counter() - t
        ex1, pop1, kc1, stp1 = ctx.astar_debug_stats(reps)
        ctx._l.sc_astar_debug_peek(ctx._h, buf)
        print("   most expanding query alone: %.2f ms, %d expansions, %d popped, in-kernel %.2f ms, overflowed %d" % (dt1 * 1e3, ex1[0], pop1[0], kc1[0] * 1024 / 2.4e6, buf[1]), flush=True)
