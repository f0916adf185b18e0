"""One-off stress: batched HIP TOPP-RA vs the CPU oracle on random plans (1..16 joints, 1..300 stages, position-dependent and
constant velocity limits, non-zero boundary velocities, some infeasible plans): statuses equal, K / x / u / t within 1e-9
where the oracle succeeds, sampled profiles within 1e-5.  python tools/toppra_stress.py [rounds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from oracle import oracle
oracle.build()
ctx = sc.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(11)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nplans = nok = 0
worst = 0.0
for r in range(rounds):
    dof = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16]))
    N = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 31, 50, 100, 200, int(rng.integers(1, 300))]))
    P = int(rng.choice([1, 3, 17, 40]))
    p0 = rng.uniform(-2, 2, (P, dof)); p1 = p0 + rng.uniform(-3, 3, (P, dof))
    v0 = rng.uniform(-1.5, 1.5, (P, dof)); v1 = rng.uniform(-1.5, 1.5, (P, dof))
    if rng.random() < 0.3:
        v0[:] = 0; v1[:] = 0
    al = rng.uniform(0.3, 5.0, (P, dof))
    alo, ahi = -al, al * rng.uniform(0.5, 1.5, (P, dof))
    if rng.random() < 0.15:
        k = int(rng.integers(P)); alo[k, 0], ahi[k, 0] = 1.0, 0.5          # an empty acceleration interval
    per_stage = rng.random() < 0.4
    if per_stage:
        s = np.arange(N + 1) / N
        vhi = rng.uniform(0.3, 2.0, (P, 1, dof)) * (0.6 + 0.4 * np.abs(np.sin(rng.uniform(1, 6) * s))[None, :, None])
    else:
        vhi = rng.uniform(0.3, 3.0, (P, dof))
    sd0, sd1 = (float(rng.uniform(0, 0.8)), float(rng.uniform(0, 0.8))) if rng.random() < 0.5 else (0.0, 0.0)
    out = ctx.toppra(t(p0), t(p1), t(v0), t(v1), t(-vhi), t(vhi), t(alo), t(ahi), N=N, sd_start=sd0, sd_end=sd1)
    smp = ctx.toppra_sample(t(p0), t(p1), t(v0), t(v1), out["x"], out["t"], 0.05, 2048)
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    sm = {k: v.cpu().numpy() for k, v in smp.items()}
    for p in range(P):
        ref = oracle.toppra(p0[p], p1[p], v0[p], v1[p], -vhi[p], vhi[p], alo[p], ahi[p], N=N, sd_start=sd0, sd_end=sd1)
        assert o["status"][p] == ref["status"], (r, p, dof, N, o["status"][p], ref["status"])
        nplans += 1
        if ref["status"] != 0:
            continue
        nok += 1
        for k in ("K", "x", "u", "t"):
            err = np.max(np.abs(o[k][p] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
            worst = max(worst, err)
            assert err < 1e-9, (r, p, dof, N, k, err)
        rs = oracle.toppra_sample(p0[p], p1[p], v0[p], v1[p], ref["x"], ref["t"], 0.05, 2048)
        n = min(rs["length"], 2048)
        if sm["length"][p] != rs["length"]:
            # ceil(T / dt) can differ when T / dt is within rounding of an integer
            assert abs(ref["t"][-1] / 0.05 - round(ref["t"][-1] / 0.05)) < 1e-6, (r, p, sm["length"][p], rs["length"])
            continue
        for k in ("pos", "vel", "acc"):
            scale = np.max(np.abs(rs[k])) + 1e-30
            assert np.max(np.abs(sm[k][p, :, :n] - rs[k][:, :n])) / scale < 1e-5, (r, p, dof, N, k)
    if r % 20 == 0:
        print("round", r, "ok", (dof, N, P, per_stage), flush=True)
print("stress ok:", nplans, "plans,", nok, "feasible; worst relative difference of K/x/u/t %.2e" % worst)
