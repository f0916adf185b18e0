"""One-off: TOPP-RA on plans scaled by 1e-9 .. 1e9 in position and 1e-6 .. 1e6 in acceleration against the oracle (the sweeps
divide through precomputed reciprocals: this looks for ranges where that differs from a division)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
rng = np.random.default_rng(5)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
worst = 0.0; nbad = 0; n = 0
for k in (-9, -6, -3, 0, 3, 6, 9):
    for ka in (-6, 0, 6):
        sc_ = 10.0 ** k; sa = 10.0 ** ka
        P, dof, N = 24, 4, 64
        p0 = rng.uniform(-2, 2, (P, dof)) * sc_; p1 = p0 + rng.uniform(0.5, 3, (P, dof)) * sc_
        v0 = rng.uniform(0.2, 1.5, (P, dof)) * sc_; v1 = rng.uniform(0.2, 1.5, (P, dof)) * sc_
        vh = rng.uniform(0.5, 3.0, (P, dof)) * sc_ * np.sqrt(sa); al = rng.uniform(0.5, 5.0, (P, dof)) * sc_ * sa
        out = ctx.toppra(t(p0), t(p1), t(v0), t(v1), t(-vh), t(vh), t(-al), t(al), N=N)
        torch.cuda.synchronize()
        o = {kk: v.cpu().numpy() for kk, v in out.items()}
        for p in range(P):
            ref = oracle.toppra(p0[p], p1[p], v0[p], v1[p], -vh[p], vh[p], -al[p], al[p], N=N)
            n += 1
            if o["status"][p] != ref["status"]:
                nbad += 1; print("status differs", k, ka, p, o["status"][p], ref["status"]); continue
            if ref["status"] == 0:
                for kk in ("K", "x", "u", "t"):
                    err = np.max(np.abs(o[kk][p] - ref[kk])) / (np.max(np.abs(ref[kk])) + 1e-300)
                    worst = max(worst, err)
print("plans", n, "status mismatches", nbad, "worst rel diff %.3e" % worst)
