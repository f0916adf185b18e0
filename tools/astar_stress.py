"""One-off stress: batched HIP A* vs the CPU oracle on many random maps (sizes, densities, clearances, map families);
any mismatch in status / cost / len / path is fatal.  python tools/astar_stress.py [rounds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
nq = nfound = nbad_ex = 0
for r in range(rounds):
    W, H = int(rng.integers(9, 400)), int(rng.integers(9, 400))
    fam = rng.integers(0, 3)
    if fam == 0:
        occ = synth.salt_grid(W, H, float(rng.uniform(0.02, 0.45)), seed=int(rng.integers(1 << 30)))
    elif fam == 1:
        occ = synth.block_grid(W, H, float(rng.uniform(0.05, 0.4)), seed=int(rng.integers(1 << 30)), smin=2, smax=max(3, min(W, H) // 4))
    else:  # maze-like: salt + walls with gaps
        occ = synth.salt_grid(W, H, 0.05, seed=int(rng.integers(1 << 30)))
        for x in range(4, W - 4, int(rng.integers(5, 17))):
            occ[1:H - 1, x] = 1
            for _ in range(2):
                y = int(rng.integers(1, H - 1)); occ[max(1, y - 1):y + 2, x] = 0
    r2 = int(rng.choice([0, 0, 1, 2, 4, 9]))
    d2 = oracle.edt(occ)
    trav = d2 >= max(r2, 1)
    if trav.sum() < 4:
        continue
    Q = 96 if r % 10 else 6000      # now and then more queries than the launch has slots: every block takes several
    free = np.flatnonzero(trav.ravel()).astype(np.int32)
    s = rng.choice(free, Q).astype(np.int32); g = rng.choice(free, Q).astype(np.int32)
    Lmax = 4 * (W + H)
    ref = oracle.astar_batch(d2, s, g, r2=r2, Lmax=Lmax, nthreads=8)
    out = ctx.astar_batch(torch.from_numpy(d2).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), r2=r2, Lmax=Lmax)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for k in ("status", "cost", "len"):
        assert np.array_equal(got[k], ref[k]), (r, W, H, fam, r2, k)
    for q in range(Q):
        if ref["status"][q] == 0:
            assert np.array_equal(got["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]]), (r, W, H, fam, r2, q)
    ex = ctx.astar_debug_stats(Q)[0]
    if not np.array_equal(ex, ref["expanded"]):
        bad = np.flatnonzero(ex != ref["expanded"])
        print("expansion counts differ: round", r, (W, H), "family", int(fam), "r2", r2, "queries", bad[:8].tolist(), "status", ref["status"][bad[:8]].tolist(),
              "gpu", ex[bad[:8]].tolist(), "oracle", ref["expanded"][bad[:8]].tolist(), "cost", ref["cost"][bad[:8]].tolist(), flush=True)
        nbad_ex += 1
    nq += Q; nfound += int((ref["status"] == 0).sum())
    if r % 25 == 0:
        print("round", r, "ok", flush=True)
print("stress ok:", nq, "queries,", nfound, "with paths;", nbad_ex, "rounds with differing expansion counts")
assert nbad_ex == 0
