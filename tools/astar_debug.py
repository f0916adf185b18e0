"""Scratch: small A* cases one by one with progress lines (bounded; for bringing a new kernel up).
usage: astar_debug.py [lib.so] [peek]   -- `peek`: poll the kernel's progress markers (library built with -DASTAR_MARKERS)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    sc.LIB_PATH = os.path.join(sc.NATIVE_DIR, sys.argv[1])
peek = "peek" in sys.argv
from sea_current_amd import synth
from oracle import oracle
ctx = sc.Context(0, use_torch_stream=False)
print("ctx ok", flush=True)
for (W, H, p, Q) in ((40, 28, 0.1, 1), (40, 28, 0.1, 8), (64, 64, 0.25, 64), (256, 256, 0.2, 64), (1024, 1024, 0.2, 96)):
    occ = synth.salt_grid(W, H, p, seed=1)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, Q, seed=1)
    ref = oracle.astar_batch(d2, s, g, Lmax=4096, nthreads=8)
    d2d, sd, gd = torch.from_numpy(d2).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
    torch.cuda.synchronize()
    print("case", W, H, p, Q, "launch", flush=True)
    t = time.perf_counter()
    out = ctx.astar_batch(d2d, sd, gd, Lmax=4096)
    print("   enqueued after %.3f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
    if peek:
        buf = (C.c_int32 * 16)()
        for i in range(6):
            time.sleep(0.3)
            ctx._l.sc_astar_debug_peek(ctx._h, buf)
            print("   peek", list(buf), flush=True)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print("   synchronised", flush=True)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    ex = ctx.astar_debug_stats(Q)[0]
    bad = [q for q in range(Q) if o["status"][q] != ref["status"][q] or o["cost"][q] != ref["cost"][q] or o["len"][q] != ref["len"][q]
           or (ref["status"][q] == 0 and not np.array_equal(o["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]))]
    print("   %.2f ms; status %s; expansions gpu %d ref %d; mismatching queries: %s" % (dt * 1e3, np.bincount(o["status"], minlength=5).tolist(),
          ex.sum(), ref["expanded"].sum(), bad[:10]), flush=True)
    if bad:
        q = bad[0]
        print("   first bad: q", q, "gpu", o["status"][q], o["cost"][q], o["len"][q], "ref", ref["status"][q], ref["cost"][q], ref["len"][q], "exp", ex[q], ref["expanded"][q], flush=True)
