"""Long stress run: batched HIP TOPP-RA vs the CPU oracle on random plans (bounded version in tests/test_gpu_stress.py).
python tools/toppra_stress.py [rounds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sea-current_amd", "python"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sea_current_amd as sc
from oracle import oracle
import stress_cases as cases
oracle.build()
ctx = sc.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(11)
nplans = nok = 0
worst = 0.0
for r in range(rounds):
    P, ok, w = cases.toppra_round(ctx, oracle, rng)
    nplans += P; nok += ok; worst = max(worst, w)
    if r % 20 == 0:
        print("round", r, "ok", flush=True)
print("stress ok:", nplans, "plans,", nok, "feasible; worst relative difference of K/x/u/t %.2e" % worst)
